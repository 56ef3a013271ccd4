// tests/wave_emu/emu_runtime.cpp — fiber scheduler + host-runtime stubs of the TEST-ONLY emulator.
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <vector>

#include "hip/hip_runtime.h"

namespace mpcq {
alignas(16) unsigned char smem_raw[160 * 1024];  // the kernels' `extern __shared__` block
}

namespace emu {
Lane* cur = nullptr;
namespace {
constexpr size_t STACK = 512 * 1024;
struct Fiber { ucontext_t ctx; Lane lane; bool done = false; std::vector<char> stack; };
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
const std::function<void()>* body_ptr = nullptr;
int current = -1;
std::vector<uint64_t> exch(1024), exch2(1024), exch3(1024);
void trampoline() {
  (*body_ptr)();
  fibers[current].done = true;
  swapcontext(&fibers[current].ctx, &sched_ctx);
}
}  // namespace
uint64_t* exchange() { return exch.data(); }
uint64_t* exchange2() { return exch2.data(); }
uint64_t* exchange3() { return exch3.data(); }
void sync() { swapcontext(&fibers[current].ctx, &sched_ctx); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  static const bool reverse = getenv("MPCQ_EMU_REVERSE") != nullptr;
  static const char* shuffle = getenv("MPCQ_EMU_SHUFFLE");   // seed: lanes are resumed in a pseudo-random order that changes at every rendezvous
  static uint64_t rng = shuffle ? (uint64_t)atoll(shuffle) * 0x9E3779B97F4A7C15ull + 1 : 0;
  std::vector<int> order(block.x);
  const int nt = block.x;
  if (shmem > sizeof(mpcq::smem_raw)) { fprintf(stderr, "emu: shared memory request too large\n"); abort(); }
  fibers.resize(nt);
  body_ptr = &body;
  static const bool poison = getenv("MPCQ_EMU_POISON") != nullptr;
  for (unsigned b = 0; b < grid.x; ++b) {
    std::fill(exch3.begin(), exch3.end(), 0);   // per-lane counts of EXEC reads (checked build)
    if (poison) std::memset(mpcq::smem_raw, 0xFF, sizeof(mpcq::smem_raw));   // LDS is not initialised on the device either: all-ones = NaN
    for (int t = 0; t < nt; ++t) {
      Fiber& f = fibers[t];
      f.done = false;
      f.lane.tid = dim3(t); f.lane.bid = dim3(b); f.lane.bdim = block; f.lane.gdim = grid;
      if (f.stack.size() != STACK) f.stack.resize(STACK);
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack.data();
      f.ctx.uc_stack.ss_size = STACK;
      f.ctx.uc_link = &sched_ctx;
      makecontext(&f.ctx, trampoline, 0);
    }
    // every pass resumes each live lane once: lanes run to their next rendezvous in turn
    bool live = true;
    while (live) {
      live = false;
      for (int k = 0; k < nt; ++k) order[k] = reverse ? nt - 1 - k : k;
      if (shuffle)
        for (int k = nt - 1; k > 0; --k) {   // Fisher-Yates with xorshift64
          rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
          std::swap(order[k], order[(int)(rng % (uint64_t)(k + 1))]);
        }
      for (int k = 0; k < nt; ++k) {
        const int t = order[k];
        if (fibers[t].done) continue;
        current = t;
        cur = &fibers[t].lane;
        swapcontext(&sched_ctx, &fibers[t].ctx);
        if (!fibers[t].done) live = true;
      }
    }
  }
  cur = nullptr;
}
}  // namespace emu

struct emuEvent { std::chrono::steady_clock::time_point t; };
const char* hipGetErrorString(hipError_t) { return "emu"; }
hipError_t hipGetDeviceCount(int* n) { *n = 8; return hipSuccess; }   // pretend an 8-GPU node so multi-rank tests can use LOCAL_RANK as the ordinal
static thread_local int emu_device = 0;
hipError_t hipSetDevice(int d) { emu_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = emu_device; return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (launches execute synchronously here)
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t lds) {   // LDS-limited, at most two waves per SIMD
  const size_t g = 1280, per = ((lds ? lds : 1) + g - 1) / g * g;
  *n = (int)std::min<size_t>(8, (160 * 1024) / per);
  return hipSuccess;
}
