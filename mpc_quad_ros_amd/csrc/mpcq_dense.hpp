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
// One 64-lane workgroup per deferred quadrotor (no workgroup barriers: one wave).  LDS: H and its factor (row stride nv + 1 doubles so
// that the lanes of a column access hit different banks), the current and next G_i, vectors.  N <= 20 (130 KB).
#pragma once
#include "mpcq_kernels.hpp"

namespace mpcq {

__host__ __device__ inline int dense_ldp(int nv) { return nv + 1; }
__host__ __device__ inline size_t dense_lds_bytes(int N) {
  const int nv = N * NU;
  return (size_t)(2 * nv * dense_ldp(nv) + 2 * NX * nv + NX * ABW + 16 * nv + 128) * sizeof(double);
}

// value of x held by lane `src` (wave-uniform src): two v_readlane
__device__ inline double lane_get(double x, int src) { return bc(x, src); }

// Cholesky factorisation of the nv x nv matrix M (lower triangle, row stride ld) in place; invd[k] = 1 / L[k][k].
// 4-column panels: the diagonal 4x4 block is factorised redundantly by every lane, each lane solves the panel rows it owns
// (rows lane, lane + 64), then the trailing 4x4 tiles are updated from the panel (one tile per lane and round).
__device__ inline bool dense_cholesky(double* M, int nv, int ld, double* invd) {
  const int tid = lane_id();
  const int nt = nv >> 2;   // 4x4 tiles per dimension (nv = 4N)
  bool ok = true;
  for (int p = 0; p < nt; ++p) {
    const int j0 = 4 * p;
    // ---- diagonal block (every lane the same arithmetic)
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
    __syncthreads();   // every lane has read the block before anyone overwrites it
    if (tid < 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c <= tid) M[(j0 + tid) * ld + j0 + c] = tid == 0 ? l[0][c] : (tid == 1 ? l[1][c] : (tid == 2 ? l[2][c] : l[3][c]));
      invd[j0 + tid] = tid == 0 ? id[0] : (tid == 1 ? id[1] : (tid == 2 ? id[2] : id[3]));
    }
    // ---- panel rows below the block: x L11' = A21 row
    for (int r = j0 + 4 + tid; r < nv; r += 64) {
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
    __syncthreads();
    // ---- trailing update: tile (ab, bb), p < bb <= ab < nt:  A[ab][bb] -= L[ab][p] L[bb][p]'
    const int rem = nt - p - 1, ntile = rem * (rem + 1) / 2;
    for (int t = tid; t < ntile; t += 64) {
      int ab = 0, acc = 0;                       // unrank t -> (ab, bb) in the lower triangle of rem x rem
      while (acc + ab + 1 <= t) { acc += ab + 1; ++ab; }
      const int bb = t - acc;
      const int ra = 4 * (p + 1 + ab), rb = 4 * (p + 1 + bb);
      double la[4][4], lb[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { la[r][c] = M[(ra + r) * ld + j0 + c]; lb[r][c] = M[(rb + r) * ld + j0 + c]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = M[(ra + r) * ld + rb + c];
#pragma unroll
          for (int k = 0; k < 4; ++k) v -= la[r][k] * lb[c][k];
          M[(ra + r) * ld + rb + c] = v;
        }
    }
    __syncthreads();
  }
  return wave_min<int>(ok ? 1 : 0) != 0;
}

// Solve L L' x = b for the factor above.  The right-hand side lives in registers: lane t holds rows t (b0) and t + 64 (b1).
__device__ inline void dense_solve(const double* M, int nv, int ld, const double* invd, double& b0, double& b1) {
  const int tid = lane_id();
  for (int k = 0; k < nv; ++k) {              // forward: y_k = b_k / L_kk ; b_r -= L_rk y_k (r > k)
    const double yk = (k < 64 ? lane_get(b0, k) : lane_get(b1, k - 64)) * invd[k];
    if (tid == k) b0 = yk;
    if (tid + 64 == k) b1 = yk;
    if (tid > k && tid < nv) b0 -= M[tid * ld + k] * yk;
    if (tid + 64 > k && tid + 64 < nv) b1 -= M[(tid + 64) * ld + k] * yk;
  }
  for (int k = nv - 1; k >= 0; --k) {         // backward: x_k = y_k / L_kk ; y_r -= L_kr x_k (r < k)
    const double xk = (k < 64 ? lane_get(b0, k) : lane_get(b1, k - 64)) * invd[k];
    if (tid == k) b0 = xk;
    if (tid + 64 == k) b1 = xk;
    if (tid < k) b0 -= M[k * ld + tid] * xk;
    if (tid + 64 < k) b1 -= M[k * ld + tid + 64] * xk;
  }
}

template <typename TQ>
__global__ void __launch_bounds__(64) dense_ipm_kernel(const DevModel<TQ> m, const DevState<TQ> st, const int par) {
  const int tid = lane_id(), N = m.N, nv = N * NU, ld = dense_ldp(nv);
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
         *invd = vec + 13 * nv;
  double* sm = vec + 16 * nv;        // d [16] | dn [16] | ev [16] | cs [16] | q [16] | qe [16]
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int b = st.defer_list[par * m.B + e];
    const TQ* G = st.stage + (size_t)b * L.gtotal;
    double* rec = st.defer_rec + (size_t)b * defer_stride(N);
    double* out = rec + 3 * nv + 16;
    // ---- condensing: free response d_i, G_i, H = sum G_i' Q_i G_i, g = sum G_i' (Q_i d_i + qv_i)
    for (int it = tid; it < NX * nv; it += 64) Gc[it] = 0.0;
    if (tid < 16) { sm[tid] = tid < NX ? rec[3 * nv + tid] : 0.0; sm[64 + tid] = tid < NX ? m.h * m.W[i2o(tid)] : 0.0; sm[80 + tid] = tid < NX ? m.We[i2o(tid)] : 0.0; }
    const int nt = nv >> 2, ntile = nt * (nt + 1) / 2;
    double acc[4][16];               // up to 4 tiles of H per lane (ntile <= 210 for N <= 20)
    int tab[4], tbb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = tid + 64 * k;
      int ab = 0, a0 = 0;
      while (a0 + ab + 1 <= t) { a0 += ab + 1; ++ab; }
      tab[k] = t < ntile ? ab : -1; tbb[k] = t - a0;
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[k][x] = 0.0;
    }
    double g0 = 0.0, g1 = 0.0;       // g rows tid, tid + 64
    __syncthreads();
    for (int i = 0; i < N; ++i) {
      for (int it = tid; it < NX * ABW; it += 64) ABs[it] = (double)G[L.AB + i * ABS + it];
      if (tid < 16) sm[48 + tid] = tid < NX ? (double)G[L.c + i * VS + tid] : 0.0;
      __syncthreads();
      for (int it = tid; it < NX * nv; it += 64) {
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
      const int ncol = 4 * (i + 1);
      if (tid < ncol) { double v = 0.0; for (int r = 0; r < NX; ++r) v += Gn[r * nv + tid] * sm[32 + r]; g0 += v; }
      if (tid + 64 < ncol) { double v = 0.0; for (int r = 0; r < NX; ++r) v += Gn[r * nv + tid + 64] * sm[32 + r]; g1 += v; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (tab[k] < 0 || tab[k] > i) continue;
        const int ca = 4 * tab[k], cb = 4 * tbb[k];
        for (int r = 0; r < NX; ++r) {
          const double q = sm[96 + r];
          double ga[4], gb[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) { ga[x] = q * Gn[r * nv + ca + x]; gb[x] = Gn[r * nv + cb + x]; }
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[k][4 * x + y] += ga[x] * gb[y];
        }
      }
      __syncthreads();
      { double* t = Gc; Gc = Gn; Gn = t; }
      if (tid < 16) sm[tid] = sm[16 + tid];
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (tab[k] < 0) continue;
      const int ca = 4 * tab[k], cb = 4 * tbb[k];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) { H[(ca + x) * ld + cb + y] = acc[k][4 * x + y]; H[(cb + y) * ld + ca + x] = acc[k][4 * x + y]; }
    }
    __syncthreads();
    for (int i = tid; i < nv; i += 64) {
      H[i * ld + i] += m.h * m.W[NX + (i & 3)];
      g[i] = (i < 64 ? g0 : g1) + rec[i];
      lb[i] = rec[nv + i]; ub[i] = rec[2 * nv + i];
    }
    __syncthreads();
    // ---- Mehrotra predictor-corrector on  min 1/2 z'Hz + g'z, lb <= z <= ub  (same start and rules as the in-kernel interior point)
    for (int i = tid; i < nv; i += 64) {
      const double w = ub[i] - lb[i];
      const double z0 = tmin(tmax(0.0, lb[i] + 0.1 * w), ub[i] - 0.1 * w);
      z[i] = z0; sl[i] = z0 - lb[i]; su[i] = ub[i] - z0;
    }
    __syncthreads();
    double gm = 1.0;
    for (int i = tid; i < nv; i += 64) { double t = g[i]; for (int j = 0; j < nv; ++j) t += H[i * ld + j] * z[j]; rd[i] = t; gm = tmax(gm, fabs(t)); }
    gm = wave_max(gm);
    for (int i = tid; i < nv; i += 64) { ll[i] = 0.1 * gm / sl[i]; lu[i] = 0.1 * gm / su[i]; }
    __syncthreads();
    const double tol = (double)m.ipm_tol;
    int it = 0, status = 2;
    for (; it < m.qp_max_iter; ++it) {
      double rdm = 0.0, mu = 0.0;
      for (int i = tid; i < nv; i += 64) {
        double t = g[i];
        for (int j = 0; j < nv; ++j) t += H[i * ld + j] * z[j];
        t += -ll[i] + lu[i];
        rd[i] = t;
        rdm = tmax(rdm, fabs(t));
        mu += sl[i] * ll[i] + su[i] * lu[i];
      }
      rdm = wave_max(rdm);
      mu = wave_sum(mu) / (2 * nv);
      if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
      if (rdm <= tol * gm && mu <= tol) { status = 0; break; }
      for (int it2 = tid; it2 < nv * nv; it2 += 64) { const int r = it2 / nv, c = it2 - r * nv; if (c <= r) M[r * ld + c] = H[r * ld + c]; }
      __syncthreads();
      for (int i = tid; i < nv; i += 64) M[i * ld + i] += ll[i] / sl[i] + lu[i] / su[i];
      __syncthreads();
      if (!dense_cholesky(M, nv, ld, invd)) { status = 4; break; }
      // predictor (sigma = 0)
      double b0 = tid < nv ? -rd[tid] - ll[tid] + lu[tid] : 0.0, b1 = tid + 64 < nv ? -rd[tid + 64] - ll[tid + 64] + lu[tid + 64] : 0.0;
      dense_solve(M, nv, ld, invd, b0, b1);
      if (tid < nv) dza[tid] = b0;
      if (tid + 64 < nv) dza[tid + 64] = b1;
      __syncthreads();
      double aff = 1.0;
      for (int i = tid; i < nv; i += 64) {
        const double d = dza[i], dl = -ll[i] - ll[i] / sl[i] * d, du = -lu[i] + lu[i] / su[i] * d;
        if (d < 0) aff = tmin(aff, -sl[i] / d);
        if (d > 0) aff = tmin(aff, su[i] / d);
        if (dl < 0) aff = tmin(aff, -ll[i] / dl);
        if (du < 0) aff = tmin(aff, -lu[i] / du);
        dll[i] = dl; dlu[i] = du;
      }
      aff = wave_min(aff);
      double mua = 0.0;
      for (int i = tid; i < nv; i += 64) mua += (sl[i] + aff * dza[i]) * (ll[i] + aff * dll[i]) + (su[i] - aff * dza[i]) * (lu[i] + aff * dlu[i]);
      mua = wave_sum(mua) / (2 * nv);
      double sigma = mua / mu;
      sigma = sigma * sigma * sigma;
      // corrector
      auto rhs = [&](int i) {
        const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i], rcu = -su[i] * lu[i] + sigma * mu + dza[i] * dlu[i];
        return -rd[i] + rcl / sl[i] - rcu / su[i];
      };
      b0 = tid < nv ? rhs(tid) : 0.0;
      b1 = tid + 64 < nv ? rhs(tid + 64) : 0.0;
      dense_solve(M, nv, ld, invd, b0, b1);
      if (tid < nv) dz[tid] = b0;
      if (tid + 64 < nv) dz[tid + 64] = b1;
      __syncthreads();
      double ap = 1.0, ad = 1.0;
      for (int i = tid; i < nv; i += 64) {
        const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i], rcu = -su[i] * lu[i] + sigma * mu + dza[i] * dlu[i];
        const double d = dz[i], dl = (rcl - ll[i] * d) / sl[i], du = (rcu + lu[i] * d) / su[i];
        if (d < 0) ap = tmin(ap, -sl[i] / d);
        if (d > 0) ap = tmin(ap, su[i] / d);
        if (dl < 0) ad = tmin(ad, -ll[i] / dl);
        if (du < 0) ad = tmin(ad, -lu[i] / du);
        dll[i] = dl; dlu[i] = du;
      }
      ap = wave_min(ap);
      ad = wave_min(ad);
      const double tau = tmax(0.995, 1.0 - mu);
      ap = tmin(1.0, tau * ap);
      ad = tmin(1.0, tau * ad);
      for (int i = tid; i < nv; i += 64) {
        z[i] += ap * dz[i]; sl[i] += ap * dz[i]; su[i] -= ap * dz[i];
        ll[i] += ad * dll[i]; lu[i] += ad * dlu[i];
      }
      __syncthreads();
    }
    for (int i = tid; i < nv; i += 64) { out[i] = z[i]; out[nv + i] = sl[i]; out[2 * nv + i] = su[i]; out[3 * nv + i] = ll[i]; out[4 * nv + i] = lu[i]; }
    if (tid == 0) { out[5 * nv] = gm; out[5 * nv + 1] = (double)it; out[5 * nv + 2] = (double)status; }
    __syncthreads();
  }
}

}  // namespace mpcq
