// Diagnostic (GPU box): time of one stage of the Riccati factorisation / of the sweeps in isolation -- 1 024 workgroups of
// one wave (one per SIMD, as in the product at B = 1 024) run the phase REPS times on zeroed data (the instruction path does
// not depend on the values).   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-strict-aliasing [-DVARIANT...] -o fs factor_stage.hip
#define MPCQ_UNROLL_FACTOR 2
#define MPCQ_UNROLL_SWEEP 10
#include "../../mpc_quad_ros_amd/csrc/mpcq_kernels.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mpcq;
typedef Cfg<double, true, 20, 10, false> C;


// timing mock of a forward sweep whose stage is ONE product with a closed-loop matrix (chain: l2g -> 4 FMA -> hsum -> add), the
// input dz_i = K_i x_i + k_i computed beside the chain; operands read like the real sweep (values meaningless)
template <bool SIDE, int PD = 2, int MODE = 0> __device__ void mock_forward(const DevModel<double>& m, double* S, double* A, const Lds& L, int dzo) {
  const int N = 20, lane = lane_id(), h = lane >> 4, c = lane & 15;
  const RMaj<double> rm(L.AB + N * ABS, L.AB, ABS, NX, h, c);
  const int j0 = (h + 2) & 3, ko0 = L.K + j0 * ABW + c;
  double xc = S[L.dx + c], qa[PD + 1][4], k0, g0, gc = S[L.Dx + c], gcn, bq[PD + 1][4];
  for (int d = 0; d < PD; ++d) { rm.load(A, d, qa[d]); for (int j = 0; j < 4; ++j) bq[d][j] = A[L.AB + d * ABS + (c < NX ? c : 0) * ABW + 10 + j]; }
  k0 = S[ko0]; g0 = S[L.vin + j0];
#pragma unroll 10
  for (int i = 0; i < N; ++i) {
    const int ip = i + 1 < N ? i + 1 : i, ig = i + PD < N ? i + PD : N - 1;
    rm.load(A, ig, qa[PD]);
    for (int j = 0; j < 4; ++j) bq[PD][j] = A[L.AB + ig * ABS + (c < NX ? c : 0) * ABW + 10 + j];
    const double k0n = S[ko0 + ip * KS], g0n = S[L.vin + ip * VS + j0];
    gcn = S[L.Dx + ip * VS + c];
    const double kv0 = S[L.vin + i * VS], kv1 = S[L.vin + i * VS + 1], kv2 = S[L.vin + i * VS + 2], kv3 = S[L.vin + i * VS + 3];
    double d = gc;
    if (SIDE) {
      d += (bq[0][0] * kv0 + bq[0][1] * kv1) + (bq[0][2] * kv2 + bq[0][3] * kv3);
      const double u0 = rowsum(k0 * xc) + g0;
      if (c == 0) S[dzo + i * NU + j0] = u0;
    }
    double xv[4];
    if (MODE == 1) { xv[0] = xv[1] = xv[2] = xv[3] = xc; } else l2g<double>(xc, h, xv);
    const double t = (qa[0][0] * xv[0] + qa[0][1] * xv[1]) + (qa[0][2] * xv[2] + qa[0][3] * xv[3]);
    double xn = (MODE == 2 ? t : hsum(t)) + d;
    xn = c < NX ? xn : 0.0;
    xc = xn;
    if (lane < VS) S[L.dx + (i + 1) * VS + lane] = xn;
    k0 = k0n; g0 = g0n; gc = gcn;
    shift<double, PD>(qa); shift<double, PD>(bq);
  }
  __syncthreads();
}

template <int WHAT> __global__ __launch_bounds__(64) void kern(DevModel<double> m, double* stage, int reps, int* sink) {
  const Lds L = lds_layout(20, 10, 1);
  double* S = reinterpret_cast<double*>(smem_raw + L.dbytes);
  double* G = stage + (size_t)blockIdx.x * L.gtotal;
  int acc = 0;
  for (int r = 0; r < reps; ++r) {
    if (WHAT == 0) acc += riccati_factor<C, false>(m, S, G, S, L) ? 1 : 0;
    if (WHAT == 1) { double g = 0; acc += riccati_factor<C, true, true>(m, S, G, S, L, &g, G + L.mrow, G + L.pst, -1) ? 1 : 0; }
    if (WHAT == 2) riccati_forward<C>(m, S, G, S, L, L.dz);
    if (WHAT == 3) riccati_backward_vec<C>(m, S, G, S, L, false);
    if (WHAT == 4) riccati_forward<C, true>(m, S, G, S, L, L.dz);
    if (WHAT == 5) { double g = 0; acc += riccati_factor<C, true, true>(m, S, G, S, L, &g, G + L.mrow, (double*)nullptr, -1) ? 1 : 0; }
    if (WHAT == 6) { double g = 0; acc += riccati_factor<C, true, true>(m, S, G, S, L, &g, (double*)nullptr, (double*)nullptr, -1) ? 1 : 0; }
    if (WHAT == 8) mock_forward<true>(m, S, G, L, L.dz);
    if (WHAT == 9) mock_forward<false>(m, S, G, L, L.dz);
    if (WHAT == 10) mock_forward<false, 6>(m, S, G, L, L.dz);
    if (WHAT == 11) mock_forward<false, 2, 1>(m, S, G, L, L.dz);
    if (WHAT == 12) mock_forward<false, 2, 2>(m, S, G, L, L.dz);
    if (WHAT == 13) mock_forward<false, 2, 0>(m, S, S, L, L.dz);
    if (WHAT == 7) acc += riccati_factor<C, true>(m, S, G, S, L) ? 1 : 0;
  }
  if (acc == 12345) sink[blockIdx.x] = acc;
}

template <int WHAT> static void run(const char* name, DevModel<double> m, double* stage, int* sink, size_t lds, int B, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<WHAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern<WHAT>, dim3(B), dim3(64), lds, 0, m, stage, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern<WHAT>, dim3(B), dim3(64), lds, 0, m, stage, reps, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipError_t e = hipGetLastError();
  printf("%-50s %8.3f us per call  = %7.0f ns per stage   (%s)\n", name, 1e3 * ms / reps, 1e6 * ms / reps / 20, hipGetErrorString(e));
}

int main() {
  const int B = 1024, reps = 200;
  const Lds L = lds_layout(20, 10, 1);
  const size_t lds = lds_bytes<double>(L);
  DevModel<double> m; memset(&m, 0, sizeof(m));
  m.N = 20; m.nb = 10; m.B = B; m.gab = 1; m.eps = 1.1e-16;
  double* stage; int* sink;
  hipMalloc(&stage, (size_t)B * L.gtotal * sizeof(double)); hipMemset(stage, 0, (size_t)B * L.gtotal * sizeof(double));
  hipMalloc(&sink, B * sizeof(int));
  // NOTE: LDS is not initialised: the working-set masks are whatever the previous kernel left, so the "active set" rows time
  // a mix of pinned and free inputs (the row stores of pinned inputs included); the instruction path of the interior-point
  // factorisation and of the sweeps does not depend on the data.
  printf("LDS per workgroup %zu bytes, global record %zu bytes\n", lds, (size_t)L.gtotal * sizeof(double));
  run<0>("factor (interior point)", m, stage, sink, lds, B, reps);
  run<1>("factor (active set, affine, store)", m, stage, sink, lds, B, reps);
  run<5>("factor (active set, affine, no P store)", m, stage, sink, lds, B, reps);
  run<6>("factor (active set, affine, no P store, no rows)", m, stage, sink, lds, B, reps);
  run<7>("factor (active set masks only)", m, stage, sink, lds, B, reps);
  run<2>("forward sweep", m, stage, sink, lds, B, reps);
  run<8>("mock: forward sweep, one product per stage", m, stage, sink, lds, B, reps);
  run<9>("mock: the same without the side computations", m, stage, sink, lds, B, reps);
  run<10>("mock chain, prefetch depth 6", m, stage, sink, lds, B, reps);
  run<11>("mock chain, no l2g", m, stage, sink, lds, B, reps);
  run<12>("mock chain, no hsum", m, stage, sink, lds, B, reps);
  run<13>("mock chain, operands from LDS", m, stage, sink, lds, B, reps);
  run<3>("backward vector recursion", m, stage, sink, lds, B, reps);
  run<4>("forward sweep (affine)", m, stage, sink, lds, B, reps);
  return 0;
}
