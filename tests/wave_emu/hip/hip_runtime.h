// tests/wave_emu/hip/hip_runtime.h — TEST-ONLY lane-level emulator of the HIP constructs the
// mpcq kernels use, so the *unchanged* product sources (mpc_quad_ros_amd/csrc/*.hip|hpp) can be
// compiled for the host and their lane logic debugged and regression-tested in a container
// without a GPU.  Each workgroup lane is a ucontext fiber; __syncthreads() and __shfl_xor() are
// cooperative rendezvous points; lanes of a phase run one after another (order selectable, so
// a missing barrier shows up as an order-dependent result).
//
// NOT part of the product: libmpcq.so is always built by hipcc for gfx950 and has no CPU path.
// The library built from this shim is named libmpcq_emu.so and is loaded only by tests that ask
// for it explicitly.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __device__
#define __host__
#define __global__
#define __shared__
#define __align__(n) alignas(n)
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Lane { dim3 tid, bid, bdim, gdim; };
extern Lane* cur;
void sync();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
uint64_t* exchange();  // one 8-byte slot per lane
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)

inline void __syncthreads() { emu::sync(); }
template <typename T> inline T __shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t* ex = emu::exchange();
  const unsigned t = emu::cur->tid.x;
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  ex[t] = raw;
  emu::sync();
  const unsigned src = (t & ~63u) | ((t ^ (unsigned)mask) & 63u);
  raw = ex[src];
  emu::sync();
  T r;
  std::memcpy(&r, &raw, sizeof(T));
  return r;
}
inline int atomicAdd(int* p, int v) { const int old = *p; *p = old + v; return old; }   // lanes run one after another
inline float __expf(float x) { return std::exp(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __double2loint(double d) { uint64_t u; std::memcpy(&u, &d, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double d) { uint64_t u; std::memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; std::memcpy(&d, &u, 8); return d; }
inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }
// EXEC participation: every emulated cross-lane operation counts itself per lane (emu::arrive) and, behind its rendezvous,
// lane 0 of the wave verifies that all 64 lanes have executed the same number of them -- a lane that skipped one under
// divergent control flow (the thing a partial EXEC mask is on the device) is behind, and the run aborts naming the operation.
// The checked build's own EXEC test (ck_exec in mpcq_kernels.hpp) therefore always sees a full mask here; it does its work on
// the GPU.
namespace emu { uint64_t* exchange3(); }
namespace emu {
inline void arrive() { exchange3()[cur->tid.x] += 1; }
inline void verify(const char* what) {
  const unsigned t = cur->tid.x;
  if ((t & 63u) != 0) return;
  const uint64_t* c = exchange3();
  for (unsigned l = 1; l < 64 && l < cur->bdim.x; ++l)
    if (c[(t & ~63u) | l] != c[t]) {
      std::fprintf(stderr, "emu: %s executed under divergent control flow: lane %u has taken part in %llu cross-lane operations, lane 0 in %llu (workgroup %u)\n",
                   what, l, (unsigned long long)c[(t & ~63u) | l], (unsigned long long)c[t], cur->bid.x);
      std::abort();
    }
}
}  // namespace emu
inline unsigned long long __builtin_amdgcn_s_getpc() { return 0; }
inline int atomicCAS(int* p, int cmp, int v) { const int old = *p; if (old == cmp) *p = v; return old; }
inline unsigned long long __builtin_amdgcn_read_exec() { return ~0ull; }
// v_readlane_b32: every lane receives lane `src`'s value (src wave-uniform)
inline int __builtin_amdgcn_readlane(int v, int src) {
  uint64_t* ex = emu::exchange();
  const unsigned t = emu::cur->tid.x;
  ex[t] = (uint32_t)v;
  emu::arrive();
  emu::sync();
  emu::verify("v_readlane");
  const int r = (int)(uint32_t)ex[(t & ~63u) | ((unsigned)src & 63u)];
  emu::sync();
  return r;
}
// v_readfirstlane_b32 with a full EXEC mask: lane 0's value
inline int __builtin_amdgcn_readfirstlane(int v) { return __builtin_amdgcn_readlane(v, 0); }
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: D = A(16x4) B(4x16) + C, lane (h = lane>>4, c = lane&15)
// holds A[c][h], B[h][c]; C/D rows: f32 4h+reg, f64 h+4reg (MI355X_MICROARCH / cdna_hip_programming.md section 3)
namespace emu { uint64_t* exchange2(); }
template <typename T, typename V4, bool F64>
inline V4 emu_mfma(T a, T b, V4 c) {
  uint64_t* ea = emu::exchange();
  uint64_t* eb = emu::exchange2();
  const unsigned t = emu::cur->tid.x, w = t & ~63u, l = t & 63u, h = l >> 4, col = l & 15;
  uint64_t ra = 0, rb = 0;
  std::memcpy(&ra, &a, sizeof(T));
  std::memcpy(&rb, &b, sizeof(T));
  ea[t] = ra; eb[t] = rb;
  emu::arrive();
  emu::sync();
  emu::verify("v_mfma");
  for (int reg = 0; reg < 4; ++reg) {
    const unsigned row = F64 ? h + 4 * reg : 4 * h + reg;
    T acc = c[reg];
    for (unsigned k = 0; k < 4; ++k) {
      T av, bv;
      std::memcpy(&av, &ea[w + 16 * k + row], sizeof(T));
      std::memcpy(&bv, &eb[w + 16 * k + col], sizeof(T));
      acc = std::fma(av, bv, acc);
    }
    c[reg] = acc;
  }
  emu::sync();
  return c;
}
typedef float emu_f4 __attribute__((ext_vector_type(4)));
typedef double emu_d4 __attribute__((ext_vector_type(4)));
inline emu_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f4 c, int, int, int) { return emu_mfma<float, emu_f4, false>(a, b, c); }
inline emu_d4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emu_d4 c, int, int, int) { return emu_mfma<double, emu_d4, true>(a, b, c); }
// DPP: quad_perm (ctrl 0x00..0xFF), row_ror:n (0x121..0x12F) and row_newbcast:n (0x150..0x15F) with all rows/banks enabled
inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int, bool) {
  uint64_t* ex = emu::exchange();
  const unsigned t = emu::cur->tid.x;
  ex[t] = (uint32_t)v;
  emu::arrive();
  emu::sync();
  emu::verify("DPP");
  unsigned src;
  if (ctrl <= 0xFF) src = (t & ~3u) | ((unsigned)(ctrl >> (2 * (t & 3))) & 3u);
  else if (ctrl >= 0x121 && ctrl <= 0x12F) src = (t & ~15u) | ((t - (unsigned)(ctrl - 0x120)) & 15u);
  else if (ctrl >= 0x150 && ctrl <= 0x15F) src = (t & ~15u) | (unsigned)(ctrl - 0x150);   // row_newbcast:n (gfx90a+)
  else { std::abort(); }
  const int r = ((row_mask >> ((t & 63u) >> 4)) & 1) ? (int)(uint32_t)ex[src] : old;   // rows outside row_mask keep `old`
  emu::sync();
  return r;
}
// 64-bit form (v_mov_b64_dpp for row_newbcast, two 32-bit moves otherwise): both halves take the same lane permutation
inline long long __builtin_amdgcn_update_dpp(long long old, long long v, int ctrl, int rm, int bm, bool bc) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)old, (int)(unsigned)v, ctrl, rm, bm, bc);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)((unsigned long long)old >> 32), (int)(unsigned)((unsigned long long)v >> 32), ctrl, rm, bm, bc);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
// v_permlane32_swap_b32 (gfx950): lanes 32..63 of `a` are exchanged with lanes 0..31 of `b`; returns {new a, new b}
typedef unsigned emu_u2 __attribute__((ext_vector_type(2)));
inline emu_u2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
  uint64_t* ea = emu::exchange();
  uint64_t* eb = emu::exchange2();
  const unsigned t = emu::cur->tid.x, w = t & ~63u, l = t & 63u;
  ea[t] = a; eb[t] = b;
  emu::arrive();
  emu::sync();
  emu::verify("v_permlane32_swap");
  emu_u2 r;
  r[0] = l < 32 ? a : (unsigned)eb[w + l - 32];
  r[1] = l < 32 ? (unsigned)ea[w + l + 32] : b;
  emu::sync();
  return r;
}
// v_permlane16_swap_b32 (gfx950): the odd 16-lane rows of `a` are exchanged with the even rows of `b`; returns {new a, new b}
inline emu_u2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
  uint64_t* ea = emu::exchange();
  uint64_t* eb = emu::exchange2();
  const unsigned t = emu::cur->tid.x, w = t & ~63u, l = t & 63u;
  ea[t] = a; eb[t] = b;
  emu::arrive();
  emu::sync();
  emu::verify("v_permlane16_swap");
  emu_u2 r;
  r[0] = (l & 16) ? (unsigned)eb[w + l - 16] : a;
  r[1] = (l & 16) ? b : (unsigned)ea[w + l + 16];
  emu::sync();
  return r;
}
template <typename T> inline T __shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t* ex = emu::exchange();
  const unsigned t = emu::cur->tid.x;
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  ex[t] = raw;
  emu::sync();
  raw = ex[(t & ~63u) | ((unsigned)src & 63u)];
  emu::sync();
  T r;
  std::memcpy(&r, &raw, sizeof(T));
  return r;
}

// ---- host runtime subset
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
constexpr unsigned hipStreamNonBlocking = 1;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
const char* hipGetErrorString(hipError_t);
hipError_t hipGetDeviceCount(int*);
hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int*);
#define hipHostMallocDefault 0
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostFree(void*);
struct hipDeviceProp_t { int multiProcessorCount; };
hipError_t hipGetDeviceProperties(hipDeviceProp_t*, int);
hipError_t hipMalloc(void**, size_t);
hipError_t hipFree(void*);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemcpy2DAsync(void*, size_t, const void*, size_t, size_t, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipEventCreate(hipEvent_t*);
constexpr unsigned hipEventDisableTiming = 2;
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipGetLastError();
hipError_t hipFuncSetAttribute(const void*, int, int);
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int*, const void*, int, size_t);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })
