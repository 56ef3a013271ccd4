// mpcq_dense.hpp — dense interior-point solve of the deferred box-QPs of a lockstep period (two-phase period, see the MODE_DEFER
// note in mpcq_kernels.hpp).
//
// Why a second formulation.  The fused step keeps the QP stage-sparse (Riccati recursion): one factorisation is a chain of 20
// dependent stages of small products on one wavefront, ~75 k cycles, and an interior-point iteration needs a factorisation and three
// to four sweeps (~140 k cycles).  That is the right shape for the warm active-set method (one factorisation on 93 % of the
// quadrotor-steps), but a quadrotor that needs the interior point holds the launch of the whole batch for ~0.5 ms.  For exactly those
// few quadrotors per period (0.7 % on the bench workload) the condensed form is the faster one: the 4N x 4N Hessian
// H = sum_i G_i' Q_i G_i + R is built once (G_i = d dx_i / d z, 4x4 register tiles), and every interior-point iteration is one
// dense Cholesky factorisation of H + Sigma in LDS (4-column panels + 4x4 register-tile trailing updates) and two triangular
// solves with the right-hand side in registers (v_readlane broadcasts): ~30 k cycles per iteration.  cond(H) ~ 2e6 is harmless in
// double; this kernel exists only for fp64 engines.  Its result is the interior point (z, slacks, multipliers) at the hand-over
// tolerance; the exact KKT point is then reached by the same active-set iterations as always, in the MODE_FINISH launch.
//
// One 256-thread workgroup (four waves, one per SIMD) per deferred quadrotor, which has a CU to itself.  LDS: H and its factor (row
// stride nv + 1 doubles so that the lanes of a column access hit different banks), the current and next G_i, vectors.  N <= 20 (130 KB).
#pragma once
#include "mpcq_kernels.hpp"

namespace mpcq {

__host__ __device__ inline int dense_ldp(int nv) { return nv + 1; }
__host__ __device__ inline size_t dense_lds_bytes(int N) {
  const int nv = N * NU;
  return (size_t)(2 * nv * dense_ldp(nv) + 2 * NX * nv + NX * ABW + 16 * nv + 144) * sizeof(double);
}

constexpr int DT = 256;   // threads of the dense kernel: four waves, one per SIMD of the CU the deferred quadrotor has to itself

// value of x held by lane `src` of the calling wave (wave-uniform src): two v_readlane
__device__ inline double lane_get(double x, int src) { return bc(x, src); }

// workgroup-wide reduction through LDS (every thread calls it; two barriers)
template <typename T, typename OP> __device__ inline T block_reduce(T v, OP op, T* scratch) {
  v = wave_reduce(v, op);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = scratch[0];
  for (int w = 1; w < DT / 64; ++w) r = op(r, scratch[w]);
  __syncthreads();
  return r;
}

// unrank t -> (ab, bb), bb <= ab, in the lower triangle enumerated row by row
__device__ inline void tri_unrank(int t, int& ab, int& bb) {
  ab = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (ab * (ab + 1) / 2 > t) --ab;
  while ((ab + 1) * (ab + 2) / 2 <= t) ++ab;
  bb = t - ab * (ab + 1) / 2;
}

// Cholesky factorisation of the nv x nv matrix M (lower triangle, row stride ld) in place; invd[k] = 1 / L[k][k].
// 4-column panels: the diagonal 4x4 block is factorised redundantly by every thread, one thread per panel row solves it against
// the block, then one thread per trailing 4x4 tile applies the rank-4 update.  Three barriers per panel.
__device__ inline bool dense_cholesky(double* M, int nv, int ld, double* invd) {
  const int tid = threadIdx.x;
  const int nt = nv >> 2;   // 4x4 tiles per dimension (nv = 4N)
  bool ok = true;
  for (int p = 0; p < nt; ++p) {
    const int j0 = 4 * p;
    double a[4][4], l[4][4], id[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) a[r][c] = M[(j0 + r) * ld + j0 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double d = a[c][c];
#pragma unroll
      for (int k = 0; k < c; ++k) d -= l[c][k] * l[c][k];
      if (!(d > 0.0)) { ok = false; d = 1.0; }
      const double sd = sqrt(d);
      l[c][c] = sd;
      id[c] = 1.0 / sd;
#pragma unroll
      for (int r = c + 1; r < 4; ++r) {
        double v = a[r][c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= l[r][k] * l[c][k];
        l[r][c] = v * id[c];
      }
    }
    __syncthreads();   // every thread has read the block before anyone overwrites it
    if (tid < 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c <= tid) M[(j0 + tid) * ld + j0 + c] = tid == 0 ? l[0][c] : (tid == 1 ? l[1][c] : (tid == 2 ? l[2][c] : l[3][c]));
      invd[j0 + tid] = tid == 0 ? id[0] : (tid == 1 ? id[1] : (tid == 2 ? id[2] : id[3]));
    }
    {   // panel rows below the block: x L11' = A21 row
      const int r = j0 + 4 + tid;
      if (r < nv) {
        double x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = M[r * ld + j0 + c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int k = 0; k < c; ++k) x[c] -= x[k] * l[c][k];
          x[c] *= id[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) M[r * ld + j0 + c] = x[c];
      }
    }
    __syncthreads();
    // trailing update: tile (ab, bb), p < bb <= ab < nt:  A[ab][bb] -= L[ab][p] L[bb][p]'
    const int rem = nt - p - 1, ntile = rem * (rem + 1) / 2;
    for (int t = tid; t < ntile; t += DT) {
      int ab, bb;
      tri_unrank(t, ab, bb);
      const int ra = 4 * (p + 1 + ab), rb = 4 * (p + 1 + bb);
      double la[4][4], lb[4][4], v[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { la[r][c] = M[(ra + r) * ld + j0 + c]; lb[r][c] = M[(rb + r) * ld + j0 + c]; v[r][c] = M[(ra + r) * ld + rb + c]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[r][c] -= la[r][k] * lb[c][k];
          M[(ra + r) * ld + rb + c] = v[r][c];
        }
    }
    __syncthreads();
  }
  return ok;   // every thread factorised the same diagonal blocks: uniform
}

// Solve L L' x = b for the factor above, redundantly in every wave (the same instruction stream on four SIMDs; no barrier).  The
// right-hand side lives in registers: lane t of a wave holds rows t (b0) and t + 64 (b1).  Four columns per LDS round trip.
__device__ inline void dense_solve(const double* M, int nv, int ld, const double* invd, double& b0, double& b1) {
  const int lane = threadIdx.x & 63, r0 = lane, r1 = lane + 64;
  for (int k0 = 0; k0 < nv; k0 += 4) {              // forward: y_k = b_k / L_kk ; b_r -= L_rk y_k (r > k)
    double l0[4], l1[4], iv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { l0[c] = r0 < nv ? M[r0 * ld + k0 + c] : 0.0; l1[c] = r1 < nv ? M[r1 * ld + k0 + c] : 0.0; iv[c] = invd[k0 + c]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c;
      const double yk = (k < 64 ? lane_get(b0, k) : lane_get(b1, k - 64)) * iv[c];
      if (r0 == k) b0 = yk;
      if (r1 == k) b1 = yk;
      if (r0 > k) b0 -= l0[c] * yk;
      if (r1 > k) b1 -= l1[c] * yk;
    }
  }
  for (int k0 = nv - 4; k0 >= 0; k0 -= 4) {         // backward: x_k = y_k / L_kk ; y_r -= L_kr x_k (r < k)
    double l0[4], l1[4], iv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { l0[c] = r0 < nv ? M[(k0 + c) * ld + r0] : 0.0; l1[c] = r1 < nv ? M[(k0 + c) * ld + r1] : 0.0; iv[c] = invd[k0 + c]; }
#pragma unroll
    for (int c = 3; c >= 0; --c) {
      const int k = k0 + c;
      const double xk = (k < 64 ? lane_get(b0, k) : lane_get(b1, k - 64)) * iv[c];
      if (r0 == k) b0 = xk;
      if (r1 == k) b1 = xk;
      if (r0 < k) b0 -= l0[c] * xk;
      if (r1 < k) b1 -= l1[c] * xk;
    }
  }
}

template <typename TQ>
__global__ void __launch_bounds__(DT) dense_ipm_kernel(const DevModel<TQ> m, const DevState<TQ> st, const int par) {
  const int tid = threadIdx.x, N = m.N, nv = N * NU, ld = dense_ldp(nv);
  const int count = st.defer_cnt[par];
  const Lds L = lds_layout(N, m.nb, 1);
  double* D = reinterpret_cast<double*>(smem_raw);
  double* H = D;
  double* M = H + nv * ld;
  double* Gc = M + nv * ld;          // [13][nv] d dx_i / d z
  double* Gn = Gc + NX * nv;
  double* ABs = Gn + NX * nv;        // [13][16] stage sensitivities
  double* vec = ABs + NX * ABW;      // 16 vectors of nv
  double *g = vec, *lb = vec + nv, *ub = vec + 2 * nv, *z = vec + 3 * nv, *sl = vec + 4 * nv, *su = vec + 5 * nv, *ll = vec + 6 * nv,
         *lu = vec + 7 * nv, *rd = vec + 8 * nv, *dza = vec + 9 * nv, *dz = vec + 10 * nv, *dll = vec + 11 * nv, *dlu = vec + 12 * nv,
         *invd = vec + 13 * nv, *part = vec + 14 * nv;   // part: [2 nv] partial sums of the matrix-vector product
  double* sm = vec + 16 * nv;        // d [16] | dn [16] | Qd+qv [16] | c [16] | stage weights [16] | terminal weights [16] | weights of the next stage [16] | reduction scratch [16]
  double* red = sm + 112;
  auto bmax = [&](double v) { return block_reduce(v, [](double a, double b) { return a > b ? a : b; }, red); };
  auto bmin = [&](double v) { return block_reduce(v, [](double a, double b) { return a < b ? a : b; }, red); };
  auto bsum = [&](double v) { return block_reduce(v, [](double a, double b) { return a + b; }, red); };
  // H z: thread (row = tid % nv, segment = tid / nv) sums its third of the row, the first nv threads add the parts
  const int segs = DT / nv >= 3 ? 3 : (DT / nv >= 2 ? 2 : 1), seglen = (nv + segs - 1) / segs;
  auto matvec = [&](const double* x, double* y_plus_g) {   // y = H x + g
    const int row = tid % nv, seg = tid / nv;
    if (seg < segs) {
      double t = 0.0;
      const int j1 = tmin(nv, (seg + 1) * seglen);
      for (int j = seg * seglen; j < j1; ++j) t += H[row * ld + j] * x[j];
      if (seg > 0) part[(seg - 1) * nv + row] = t; else y_plus_g[row] = t + g[row];
    }
    __syncthreads();
    if (tid < nv) { double t = y_plus_g[tid]; for (int sgi = 1; sgi < segs; ++sgi) t += part[(sgi - 1) * nv + tid]; y_plus_g[tid] = t; }
    __syncthreads();
  };
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int b = st.defer_list[par * m.B + e];
    const TQ* G = st.stage + (size_t)b * L.gtotal;
    double* rec = st.defer_rec + (size_t)b * defer_stride(N);
    double* out = rec + 3 * nv + 16;
    // ---- condensing: free response d_i, G_i, H = sum G_i' Q_i G_i, g = sum G_i' (Q_i d_i + qv_i)
    for (int it = tid; it < NX * nv; it += DT) Gc[it] = 0.0;
    if (tid < 16) { sm[tid] = tid < NX ? rec[3 * nv + tid] : 0.0; sm[64 + tid] = tid < NX ? m.h * m.W[i2o(tid)] : 0.0; sm[80 + tid] = tid < NX ? m.We[i2o(tid)] : 0.0; }
    const int nt = nv >> 2, ntile = nt * (nt + 1) / 2;   // <= 210 tiles of H for N <= 20: one per thread
    int tab = -1, tbb = 0;
    if (tid < ntile) tri_unrank(tid, tab, tbb);
    double acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = 0.0;
    double gacc = 0.0;               // g row tid
    __syncthreads();
    for (int i = 0; i < N; ++i) {
      for (int it = tid; it < NX * ABW; it += DT) ABs[it] = (double)G[L.AB + i * ABS + it];
      if (tid < 16) sm[48 + tid] = tid < NX ? (double)G[L.c + i * VS + tid] : 0.0;
      __syncthreads();
      for (int it = tid; it < NX * nv; it += DT) {
        const int r = it / nv, col = it - r * nv;
        double v = 0.0;
        if (col < 4 * i) {
#pragma unroll
          for (int k = 0; k < 10; ++k) v += ABs[r * ABW + k] * Gc[k * nv + col];
          if (r >= 10) v += Gc[r * nv + col];
        } else if (col < 4 * i + 4) v = ABs[r * ABW + 10 + col - 4 * i];
        Gn[it] = v;
      }
      if (tid < NX) {
        double v = sm[48 + tid] + (tid >= 10 ? sm[tid] : 0.0);
#pragma unroll
        for (int k = 0; k < 10; ++k) v += ABs[tid * ABW + k] * sm[k];
        sm[16 + tid] = v;
        const double q = i + 1 < N ? sm[64 + tid] : sm[80 + tid];
        sm[96 + tid] = q;                                                   // weights of stage i + 1
        sm[32 + tid] = q * v + (double)G[L.qv + (i + 1) * VS + tid];        // Q d + qv
      }
      __syncthreads();
      if (tid < 4 * (i + 1)) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) v += Gn[r * nv + tid] * sm[32 + r];
        gacc += v;
      }
      if (tab >= 0 && tab <= i) {
        const int ca = 4 * tab, cb = 4 * tbb;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          const double q = sm[96 + r];
          double ga[4], gb[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) { ga[x] = q * Gn[r * nv + ca + x]; gb[x] = Gn[r * nv + cb + x]; }
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[4 * x + y] += ga[x] * gb[y];
        }
      }
      __syncthreads();
      { double* t = Gc; Gc = Gn; Gn = t; }
      if (tid < 16) sm[tid] = sm[16 + tid];
      __syncthreads();
    }
    if (tab >= 0) {
      const int ca = 4 * tab, cb = 4 * tbb;
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) { H[(ca + x) * ld + cb + y] = acc[4 * x + y]; H[(cb + y) * ld + ca + x] = acc[4 * x + y]; }
    }
    __syncthreads();
    if (tid < nv) {
      H[tid * ld + tid] += m.h * m.W[NX + (tid & 3)];
      g[tid] = gacc + rec[tid];
      lb[tid] = rec[nv + tid]; ub[tid] = rec[2 * nv + tid];
      const double w = ub[tid] - lb[tid];
      const double z0 = tmin(tmax(0.0, lb[tid] + (double)m.ipm_margin * w), ub[tid] - (double)m.ipm_margin * w);
      z[tid] = z0; sl[tid] = z0 - lb[tid]; su[tid] = ub[tid] - z0;
    }
    __syncthreads();
    // ---- Mehrotra predictor-corrector on  min 1/2 z'Hz + g'z, lb <= z <= ub  (same start and rules as the in-kernel interior point)
    matvec(z, rd);
    const double gm = bmax(tid < nv ? tmax(1.0, fabs(rd[tid])) : 1.0);
    if (tid < nv) { ll[tid] = (double)m.ipm_mu0 * gm / sl[tid]; lu[tid] = (double)m.ipm_mu0 * gm / su[tid]; }
    __syncthreads();
    const double tol = (double)m.ipm_tol;
    int it = 0, status = 2;
    for (; it < m.qp_max_iter; ++it) {
      matvec(z, rd);
      if (tid < nv) rd[tid] += -ll[tid] + lu[tid];
      __syncthreads();
      const double rdm = bmax(tid < nv ? fabs(rd[tid]) : 0.0);
      const double mu = bsum(tid < nv ? sl[tid] * ll[tid] + su[tid] * lu[tid] : 0.0) / (2 * nv);
      if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
      if (rdm <= tol * gm && mu <= tol) { status = 0; break; }
      for (int it2 = tid; it2 < nv * nv; it2 += DT) { const int r = it2 / nv, c = it2 - r * nv; if (c <= r) M[r * ld + c] = H[r * ld + c]; }
      __syncthreads();
      if (tid < nv) M[tid * ld + tid] += ll[tid] / sl[tid] + lu[tid] / su[tid];
      __syncthreads();
      if (!dense_cholesky(M, nv, ld, invd)) { status = 4; break; }
      // predictor (sigma = 0)
      const int r0 = tid & 63, r1 = r0 + 64;
      double b0 = r0 < nv ? -rd[r0] - ll[r0] + lu[r0] : 0.0, b1 = r1 < nv ? -rd[r1] - ll[r1] + lu[r1] : 0.0;
      dense_solve(M, nv, ld, invd, b0, b1);
      if (tid < 64) { if (r0 < nv) dza[r0] = b0; if (r1 < nv) dza[r1] = b1; }
      __syncthreads();
      double aff = 1.0;
      if (tid < nv) {
        const int i = tid;
        const double d = dza[i], dl = -ll[i] - ll[i] / sl[i] * d, du = -lu[i] + lu[i] / su[i] * d;
        if (d < 0) aff = tmin(aff, -sl[i] / d);
        if (d > 0) aff = tmin(aff, su[i] / d);
        if (dl < 0) aff = tmin(aff, -ll[i] / dl);
        if (du < 0) aff = tmin(aff, -lu[i] / du);
        dll[i] = dl; dlu[i] = du;
      }
      aff = bmin(aff);
      const double mua = bsum(tid < nv ? (sl[tid] + aff * dza[tid]) * (ll[tid] + aff * dll[tid]) + (su[tid] - aff * dza[tid]) * (lu[tid] + aff * dlu[tid]) : 0.0) / (2 * nv);
      double sigma = mua / mu;
      sigma = sigma * sigma * sigma;
      // corrector
      auto rhs = [&](int i) {
        const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i], rcu = -su[i] * lu[i] + sigma * mu + dza[i] * dlu[i];
        return -rd[i] + rcl / sl[i] - rcu / su[i];
      };
      b0 = r0 < nv ? rhs(r0) : 0.0;
      b1 = r1 < nv ? rhs(r1) : 0.0;
      dense_solve(M, nv, ld, invd, b0, b1);
      if (tid < 64) { if (r0 < nv) dz[r0] = b0; if (r1 < nv) dz[r1] = b1; }
      __syncthreads();
      double ap = 1.0, ad = 1.0, dl_n = 0.0, du_n = 0.0;
      if (tid < nv) {
        const int i = tid;
        const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i], rcu = -su[i] * lu[i] + sigma * mu + dza[i] * dlu[i];
        const double d = dz[i];
        dl_n = (rcl - ll[i] * d) / sl[i]; du_n = (rcu + lu[i] * d) / su[i];
        if (d < 0) ap = tmin(ap, -sl[i] / d);
        if (d > 0) ap = tmin(ap, su[i] / d);
        if (dl_n < 0) ad = tmin(ad, -ll[i] / dl_n);
        if (du_n < 0) ad = tmin(ad, -lu[i] / du_n);
      }
      ap = bmin(ap);
      ad = bmin(ad);
      const double tau = tmax(0.995, 1.0 - mu);
      ap = tmin(1.0, tau * ap);
      ad = tmin(1.0, tau * ad);
      if (tid < nv) {
        z[tid] += ap * dz[tid]; sl[tid] += ap * dz[tid]; su[tid] -= ap * dz[tid];
        ll[tid] += ad * dl_n; lu[tid] += ad * du_n;
      }
      __syncthreads();
    }
    if (tid < nv) { out[tid] = z[tid]; out[nv + tid] = sl[tid]; out[2 * nv + tid] = su[tid]; out[3 * nv + tid] = ll[tid]; out[4 * nv + tid] = lu[tid]; }
    if (tid == 0) { out[5 * nv] = gm; out[5 * nv + 1] = (double)it; out[5 * nv + 2] = (double)status; }
    __syncthreads();
  }
}

}  // namespace mpcq
