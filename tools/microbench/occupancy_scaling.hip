// Diagnostic (GPU box): how the phases of the step scale with the number of resident wavefronts per CU (4 = one per SIMD,
// 6, 8 = two per SIMD).  Workgroups of one wave run ONE phase `reps` times on zeroed data (the instruction path of these
// phases does not depend on the values); residency is set through the dynamic-LDS request of the launch.  The compact-layout
// instance (<= 256 registers) is used throughout, so the code is the same at every residency.  For 8 per CU the LDS arrays
// the phase touches are packed into 20 KB by hand (the real layout of a whole step does not fit that yet -- this measures
// whether it would pay).
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-strict-aliasing -mllvm -amdgpu-mfma-vgpr-form=1 -o occ occupancy_scaling.hip
#define MPCQ_UNROLL_FACTOR 2
#define MPCQ_UNROLL_SWEEP 10
#include "../../mpc_quad_ros_amd/csrc/mpcq_kernels.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mpcq;
typedef Cfg<double, true, 20, 10, false, true> C;

template <int WHAT> __global__ __launch_bounds__(64, 2) void kern(DevModel<double> m, Lds L, double* stage, int reps, int* sink) {
  double* D = reinterpret_cast<double*>(smem_raw);
  double* S = reinterpret_cast<double*>(smem_raw + L.dbytes);
  double* G = stage + (size_t)blockIdx.x * L.gtotal;
  int acc = 0;
  for (int r = 0; r < reps; ++r) {
    if (WHAT == 0) acc += riccati_factor<C, false>(m, S, G, G, L) ? 1 : 0;
    if (WHAT == 1) riccati_forward<C>(m, S, G, G, L, L.dz);
    if (WHAT == 2) riccati_backward_vec<C>(m, S, G, G, L, false);
    if (WHAT == 3) { shoot_states<C>(m, D, S, G, L, true); __syncthreads(); shoot_sens<C>(m, S, G, L); __syncthreads(); }
  }
  if (acc == 12345) sink[blockIdx.x] = acc;
}

template <int WHAT> static void run(const char* name, DevModel<double> m, const Lds& L, double* stage, int* sink, size_t lds, int per_cu, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<WHAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nblk = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(&kern<WHAT>), 64, lds);
  const int B = 256 * per_cu * 4;   // four rounds of workgroups at this residency
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern<WHAT>, dim3(B), dim3(64), lds, 0, m, L, stage, 1, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern<WHAT>, dim3(B), dim3(64), lds, 0, m, L, stage, reps, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipError_t e = hipGetLastError();
  printf("%-34s LDS %6zu B  resident/CU %d (API %d)  %9.1f phase executions per ms   (%s)\n", name, lds, per_cu, nblk, (double)B * reps / ms, hipGetErrorString(e));
}

int main() {
  const int reps = 20;
  Lds L = lds_layout(20, 10, 2);
  const size_t full = lds_bytes<double>(L);
  DevModel<double> m; memset(&m, 0, sizeof(m));
  m.N = 20; m.nb = 10; m.gab = 2; m.eps = 1.1e-16; m.h = 0.05; m.mass = 1; m.imass = 1; m.tmax = 1; m.g = 9.81;
  for (int i = 0; i < 3; ++i) { m.J[i] = m.iJ[i] = 1; m.L2inv[i] = 1; m.sf2[i] = 1; }
  const int Bmax = 256 * 8 * 4;
  double* stage; int* sink;
  hipMalloc(&stage, (size_t)Bmax * L.gtotal * sizeof(double)); hipMemset(stage, 0, (size_t)Bmax * L.gtotal * sizeof(double));
  hipMalloc(&sink, Bmax * sizeof(int));
  printf("compact layout: %zu B LDS per workgroup, global record %zu B\n", full, (size_t)L.gtotal * sizeof(double));
  // the QP arrays the factorisation and the sweeps touch, packed (for the 8-per-CU point): offsets in doubles from S
  Lds Lp = L;
  {
    int o = 0; auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    Lp.dbytes = 0;
    Lp.zb = take(16); Lp.wq = take(48);
    Lp.ll = take(80); Lp.sl = take(80); Lp.lu = take(80); Lp.su = take(80); Lp.rt = take(80); Lp.rho = take(80); Lp.act = take(80); Lp.dz = take(80);
    Lp.vin = take(320); Lp.dx = take(336); Lp.Dx = take(336); Lp.sF = take(64); Lp.sT = take(64); Lp.stv = take(32);
    printf("packed QP arrays: %d B\n", o * 8);
  }
  const size_t res4 = 40960, res6 = 26880, res8 = 20480;
  run<0>("factorisation (interior point)", m, L, stage, sink, res4, 4, reps);
  run<0>("factorisation (interior point)", m, L, stage, sink, res6, 6, reps);
  run<0>("factorisation (interior point)", m, Lp, stage, sink, res8, 8, reps);
  run<1>("forward sweep", m, L, stage, sink, res4, 4, reps);
  run<1>("forward sweep", m, L, stage, sink, res6, 6, reps);
  run<1>("forward sweep", m, Lp, stage, sink, res8, 8, reps);
  run<2>("backward vector recursion", m, L, stage, sink, res4, 4, reps);
  run<2>("backward vector recursion", m, L, stage, sink, res6, 6, reps);
  run<2>("backward vector recursion", m, Lp, stage, sink, res8, 8, reps);
  run<3>("shooting (states + sensitivities)", m, L, stage, sink, res4, 4, reps);
  run<3>("shooting (states + sensitivities)", m, L, stage, sink, res6, 6, reps);
  return 0;
}
