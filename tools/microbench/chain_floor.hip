// Diagnostic (GPU box): the latency floor of the chains that bound a lockstep launch -- the dependent instruction chain of ONE
// stage of the Riccati factorisation and of ONE stage of a vector sweep, with everything that is not on the chain taken away:
// operands and results stay in registers (no LDS hand-over, no global loads or stores, no masks, no stores of K / Lambda^-1 /
// cost-to-go), one wavefront per SIMD.  What remains per factorisation stage is exactly the arithmetic dependency of
// riccati_factor (mpcq_kernels.hpp): 4 + 4 chained v_mfma_f64_16x16x4 (T1'' = P AB'', F'' = AB''^T [T1''|p]), ten v_readlane
// pairs of the stage Hessian, the 4x4 LDL^T with four reciprocal pivots in series (v_rcp_f64 + one Newton step), the two
// substitutions, the right-hand side dot product, the k=4 tile of the P update and the DPP return of p to column 14; per
// sweep stage: l2g (two conditional rotates + four row broadcasts), four FMAs, the permlane row sum.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-strict-aliasing -mllvm -amdgpu-mfma-vgpr-form=1 -o chain_floor chain_floor.hip
#include "../../mpc_quad_ros_amd/csrc/mpcq_kernels.hpp"
#include <cstdio>
using namespace mpcq;

template <typename TQ> __global__ __launch_bounds__(64) void factor_chain(double* out, int stages) {
  const int lane = lane_id(), h = lane >> 4, c = lane & 15;
  const bool vl = c == 14;
  TQ Pop[4], cur[4], pv[4], qdg[4];
  for (int s = 0; s < 4; ++s) { Pop[s] = TQ(1e-3) * (lane + s); cur[s] = TQ(1e-2) * (c - h + s); pv[s] = TQ(1e-3) * s; qdg[s] = (RI<TQ>(s, h) == c) ? TQ(1) : TQ(0); }
  for (int i = 0; i < stages; ++i) {
    TQ acc1[4] = {0, 0, 0, 0}, acc2[4] = {0, 0, 0, 0};
    for (int s = 0; s < 4; ++s) mfma(acc1, Pop[s], cur[s]);
    TQ b2[4];
    for (int s = 0; s < 4; ++s) b2[s] = vl ? pv[s] + acc1[s] : acc1[s];
    for (int s = 0; s < 4; ++s) mfma(acc2, cur[s], b2[s]);
    TQ Lm[4][4];
    for (int a = 0; a < 4; ++a)
      for (int q = 0; q <= a; ++q) Lm[a][q] = bc(acc2[in_s<TQ>(a)], 16 * in_h<TQ>(a) + 10 + q);
    for (int a = 0; a < 4; ++a) Lm[a][a] += TQ(1);
    TQ id[4], cm[4][4];
    for (int cc = 0; cc < 4; ++cc) {
      TQ d = Lm[cc][cc];
      for (int k = 0; k < cc; ++k) d -= Lm[cc][k] * cm[cc][k];
      d = d > TQ(0) ? d : TQ(1);
      id[cc] = trcp1(d);
      for (int a = cc + 1; a < 4; ++a) {
        TQ s2 = Lm[a][cc];
        for (int k = 0; k < cc; ++k) s2 -= Lm[a][k] * cm[cc][k];
        cm[a][cc] = s2;
        Lm[a][cc] = s2 * id[cc];
      }
    }
    TQ y[4];
    for (int j = 0; j < 4; ++j) y[j] = acc2[j];          // (the real code reads M[:, c] from the LDS hand-over: same dependency on the tile products)
    for (int cc = 1; cc < 4; ++cc)
      for (int k = 0; k < cc; ++k) y[cc] -= Lm[cc][k] * y[k];
    for (int cc = 0; cc < 4; ++cc) y[cc] *= id[cc];
    for (int cc = 2; cc >= 0; --cc)
      for (int k = cc + 1; k < 4; ++k) y[cc] -= Lm[k][cc] * y[k];
    const TQ yh = h == 0 ? y[0] : (h == 1 ? y[1] : (h == 2 ? y[2] : y[3]));
    const TQ mh = h == 0 ? acc2[0] : (h == 1 ? acc2[1] : (h == 2 ? acc2[2] : acc2[3]));
    const TQ kk = c < NX ? -yh : TQ(0), mop = c < NX ? mh : TQ(0);
    TQ dot = 0;
    for (int j = 0; j < 4; ++j) dot += y[j] * acc1[j];
    const TQ pcol = c < NX ? acc1[0] - dot : TQ(0);
    TQ C4[4];
    for (int s = 0; s < 4; ++s) C4[s] = TQ(0.5) * acc2[s] + TQ(0.25) * acc1[s] + TQ(0.125) * Pop[s] + qdg[s];
    mfma(C4, mop, kk);
    for (int s = 0; s < 4; ++s) Pop[s] = TQ(1e-3) * C4[s];      // (scaled: keeps the mock values bounded over many stages)
    l2g<TQ>(TQ(1e-3) * pcol, h, pv);
  }
  out[blockIdx.x * 64 + lane] = Pop[0] + pv[1];
}

template <typename TQ> __global__ __launch_bounds__(64) void sweep_chain(double* out, int stages) {
  const int lane = lane_id(), h = lane >> 4, c = lane & 15;
  TQ qa[4], xc = TQ(1e-3) * c;
  for (int s = 0; s < 4; ++s) qa[s] = TQ(1e-2) * (c - h + s);
  for (int i = 0; i < stages; ++i) {
    TQ xv[4];
    l2g<TQ>(xc, h, xv);
    const TQ ta = (qa[0] * xv[0] + qa[1] * xv[1]) + (qa[2] * xv[2] + qa[3] * xv[3]);
    xc = hsum(ta) + TQ(1e-3);
  }
  out[blockIdx.x * 64 + lane] = xc;
}

template <typename K> static double run(K k, double* out, int B, int stages) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, out, stages);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  return 1e6 * ms / stages;   // ns per stage
}

int main() {
  const int B = 1024, stages = 20000;   // one wavefront per SIMD, as a lockstep launch at B = 1024
  double* out; (void)hipMalloc(&out, (size_t)B * 64 * sizeof(double));
  const double f = run(factor_chain<double>, out, B, stages), s = run(sweep_chain<double>, out, B, stages);
  // float: the interior point of the fp64 instances' fallback runs in float (ipm_float_stage): v_mfma_f32_16x16x4_f32, v_rcp_f32, 32-bit DPP
  const double ff = run(factor_chain<float>, out, B, stages), sf = run(sweep_chain<float>, out, B, stages);
  printf("{\"factor_stage_chain_ns\": %.1f, \"sweep_stage_chain_ns\": %.1f, \"factor_stage_chain_f32_ns\": %.1f, \"sweep_stage_chain_f32_ns\": %.1f, "
         "\"workgroups\": %d, \"stages\": %d, "
         "\"note\": \"dependent chain only, registers only, one wavefront per SIMD (tools/microbench/chain_floor.hip)\"}\n", f, s, ff, sf, B, stages);
  return 0;
}
