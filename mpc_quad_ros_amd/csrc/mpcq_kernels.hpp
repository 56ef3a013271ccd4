// mpcq_kernels.hpp — device code of the batched MPC+RGP control step (gfx950 / CDNA4).
//
// One quadrotor per workgroup (default one 64-lane wavefront; NT = blockDim.x lanes cooperate
// through LDS).  All per-instance working data (iterate, shooting sensitivities [A_i|B_i],
// Riccati gains, IPM vectors, RGP covariance) is staged in LDS; HBM is touched once per step to
// load and once to store the persistent state, with lane-contiguous (coalesced) records.
//
// Algorithm (same mathematical step as the reference's acados SQP-RTI call, restated in
// SURVEY App. A; not a translation of acados/HPIPM code):
//   1. multiple shooting: explicit RK4 (1 step) with forward sensitivities per interval
//   2. the box-constrained QP in du is solved by a Mehrotra predictor-corrector IPM whose
//      Newton systems are solved with a Riccati recursion on the stage-sparse problem
//      (O(N) memory, well conditioned; the condensed Hessian is never formed)
//   3. full step, cost, nominal RK4 prediction, body-frame drag estimate, 3 scalar RGP updates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcq {

constexpr int NX = 13, NU = 4, NY = 17, NAB = NX * NY;
constexpr int SUBW = 40;  // per (stage, RK substage) record: x_s(13) Jvq(12) Jvv(9) Rz(3) pad

enum : int { MODE_TRAJ = 1, MODE_POST = 2 };

template <typename T>
struct DevModel {
  int N, nb, skip, Tmax, B, qp_max_iter, polish_max;
  T h, dt_pred;
  T mass, J[3], tmax, xf[4], yf[4], zl[4], g;
  T W[NY], We[NX], ulb[NU], uub[NU], uref[NU];
  T qp_tol;    // final KKT tolerance (IPM-only fallback)
  T ipm_tol;   // IPM -> active-set polish hand-over tolerance
  T eps;       // unit roundoff scale of T used for KKT sign / bound tests
  T L2inv[3], sf2[3], sn2[3];
  T rotor_drag[3], aero_drag;
  const T* basis;  // [3*nb]
  const T* Kxinv;  // [3*nb*nb]
};

template <typename T>
struct DevState {
  T* X;        // [B][(N+1)*13]
  T* U;        // [B][N*4]
  T* mu;       // [B][3*nb]
  T* C;        // [B][3*nb*nb]
  T* xpp;      // [B][13]   x_pred of the previous step
  T* yref;     // [B][N*17]
  T* yrefN;    // [B][13]
  const T* traj;  // [B][Tmax][13]
  const T* x_meas;  // [B][13]
  T* w;        // [B][4]
  T* xpred;    // [B][13]
  T* cost;     // [B]
  T* stats;    // [B][4]
  int* has_prev;
  int* idx;
  const int* tlen;
  int* status;
  int* qp_iter;
};

// ------------------------------------------------------------------ LDS layout (units of T)
struct Lds {
  int X, U, yref, x0, AB, c, alpha, basis, z, sl, su, ll, lu, grad, dza, dz, rho, kv, act, dx, Dx;
  int pv, piv, tv, K, Linv, P, T1, F, sub, rgp, red, total;
};
__host__ __device__ inline int al4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline Lds lds_layout(int N, int nb) {
  Lds L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += al4(n); return r; };
  L.X = take((N + 1) * NX);
  L.U = take(N * NU);
  L.yref = take(N * NY + NX);
  L.x0 = take(NX);
  L.AB = take(N * NAB);
  L.c = take(N * NX);
  L.alpha = take(3 * nb);
  L.basis = take(3 * nb);
  L.red = take(64);
  const int u0 = o;  // ---- union: shooting records | IPM workspace | RGP workspace
  L.sub = u0;
  const int sub_end = u0 + al4(N * 4 * SUBW);
  o = u0;
  const int nv = N * NU;
  L.z = take(nv); L.sl = take(nv); L.su = take(nv); L.ll = take(nv); L.lu = take(nv);
  L.grad = take(nv); L.dza = take(nv); L.dz = take(nv); L.rho = take(nv); L.kv = take(nv); L.act = take(nv);
  L.dx = take((N + 1) * NX);
  L.Dx = take((N + 1) * NX);
  L.pv = take(2 * NX); L.piv = take(2 * NX); L.tv = take(2 * NY + 8);
  L.K = take(N * NU * NX);
  L.Linv = take(N * 16);
  L.P = take(2 * NX * NX);
  L.T1 = take(NAB);
  L.F = take(NY * NY);
  const int ipm_end = o;
  L.rgp = u0;
  const int rgp_end = u0 + al4(3 * nb * nb) + 5 * al4(3 * nb) + 32;
  o = sub_end > ipm_end ? sub_end : ipm_end;
  if (rgp_end > o) o = rgp_end;
  L.total = o;
  return L;
}

// ------------------------------------------------------------------ small helpers
template <typename T> __device__ inline T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T> __device__ inline T wave_max(T v) {
  for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = v > w ? v : w; }
  return v;
}
template <typename T> __device__ inline T wave_min(T v) {
  for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = v < w ? v : w; }
  return v;
}
// block-wide reductions (NT may be a multiple of 64); `red` holds one slot per wave
template <typename T, int OP> __device__ inline T block_reduce(T v, T* red) {
  v = OP == 0 ? wave_sum(v) : (OP == 1 ? wave_max(v) : wave_min(v));
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = red[0];
  for (int w = 1; w < nw; ++w) { T t = red[w]; r = OP == 0 ? r + t : (OP == 1 ? (r > t ? r : t) : (r < t ? r : t)); }
  return r;
}
template <typename T> __device__ inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ inline T tmax(T a, T b) { return a > b ? a : b; }
__device__ inline float  texp(float x)  { return __expf(x); }
__device__ inline double texp(double x) { return exp(x); }
__device__ inline float  tsqrt(float x)  { return sqrtf(x); }
__device__ inline double tsqrt(double x) { return sqrt(x); }
__device__ inline float  tabs(float x)  { return fabsf(x); }
__device__ inline double tabs(double x) { return fabs(x); }

template <typename T> __device__ inline void rotmat(const T* q, T* R) {
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz);     R[2] = 2 * (qx * qz + qw * qy);
  R[3] = 2 * (qx * qy + qw * qz);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
  R[6] = 2 * (qx * qz - qw * qy);     R[7] = 2 * (qy * qz + qw * qx);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

// f(x,u) of the OCP model (src/quad_opt.py:186-251 in the reference); when `sub` != nullptr also
// writes the record the sensitivity pass needs: x(13) | d vdot/dq (3x4) | d vdot/dv (3x3) | R[:,2].
template <typename T>
__device__ inline void model_eval(const DevModel<T>& m, const T* x, const T* u, const T* alpha, const T* basis,
                                  T* f, T* sub) {
  const T* q = x + 3; const T* v = x + 7; const T* r = x + 10;
  T R[9];
  rotmat(q, R);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  f[3] = T(0.5) * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = T(0.5) * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = T(0.5) * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = T(0.5) * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  const T aT = m.tmax * (u[0] + u[1] + u[2] + u[3]) / m.mass;
  f[7] = R[2] * aT; f[8] = R[5] * aT; f[9] = R[8] * aT - m.g;
  T ty = 0, tx = 0, tz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ty += u[j] * m.yf[j]; tx += u[j] * m.xf[j]; tz += u[j] * m.zl[j]; }
  f[10] = (m.tmax * ty + (m.J[1] - m.J[2]) * r[1] * r[2]) / m.J[0];
  f[11] = (-m.tmax * tx + (m.J[2] - m.J[0]) * r[2] * r[0]) / m.J[1];
  f[12] = (m.tmax * tz + (m.J[0] - m.J[1]) * r[0] * r[1]) / m.J[2];
  T mg[3] = {0, 0, 0}, mp[3] = {0, 0, 0};
  const bool gp = alpha != nullptr;
  if (gp) {
    T vb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
    for (int d = 0; d < 3; ++d) {
      T s0 = 0, s1 = 0;
      for (int j = 0; j < m.nb; ++j) {
        const T dlt = vb[d] - basis[d * m.nb + j];
        const T k = alpha[d * m.nb + j] * m.sf2[d] * texp(T(-0.5) * dlt * dlt * m.L2inv[d]);
        s0 += k;
        s1 -= k * dlt;
      }
      mg[d] = s0;
      mp[d] = s1 * m.L2inv[d];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) f[7 + i] += R[3 * i] * mg[0] + R[3 * i + 1] * mg[1] + R[3 * i + 2] * mg[2];
  }
  if (!sub) return;
#pragma unroll
  for (int i = 0; i < NX; ++i) sub[i] = x[i];
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  // dR/dq_i, row-major 3x3 each, factor 2 applied below
  const T dR[4][9] = {{0, -qz, qy, qz, 0, -qx, -qy, qx, 0},
                      {0, qy, qz, qy, -2 * qx, -qw, qz, qw, -2 * qx},
                      {-2 * qy, qx, qw, qx, 0, qz, -qw, qz, -2 * qy},
                      {-2 * qz, -qw, qx, qw, -2 * qz, qy, qx, qy, 0}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T dvb[3] = {0, 0, 0};
    if (gp) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dvb[c] = mp[c] * (dR[i][c] * v[0] + dR[i][3 + c] * v[1] + dR[i][6 + c] * v[2]);
    }
#pragma unroll
    for (int row = 0; row < 3; ++row) {
      T val = aT * dR[i][3 * row + 2];
      if (gp) {
        val += dR[i][3 * row] * mg[0] + dR[i][3 * row + 1] * mg[1] + dR[i][3 * row + 2] * mg[2];
        val += R[3 * row] * dvb[0] + R[3 * row + 1] * dvb[1] + R[3 * row + 2] * dvb[2];
      }
      sub[13 + row * 4 + i] = 2 * val;
    }
  }
#pragma unroll
  for (int row = 0; row < 3; ++row)
#pragma unroll
    for (int col = 0; col < 3; ++col)
      sub[25 + row * 3 + col] = gp ? (R[3 * row] * mp[0] * R[3 * col] + R[3 * row + 1] * mp[1] * R[3 * col + 1] +
                                      R[3 * row + 2] * mp[2] * R[3 * col + 2])
                                   : T(0);
  sub[34] = R[2]; sub[35] = R[5]; sub[36] = R[8];
}

// one RK4 step of the model, no sensitivities (nominal prediction / plant-free uses)
template <typename T>
__device__ inline void rk4_step(const DevModel<T>& m, const T* x, const T* u, const T* alpha, const T* basis, T dt, T* xo) {
  T k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
  model_eval(m, x, u, alpha, basis, k1, (T*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k1[i];
  model_eval(m, xt, u, alpha, basis, k2, (T*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k2[i];
  model_eval(m, xt, u, alpha, basis, k3, (T*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
  model_eval(m, xt, u, alpha, basis, k4, (T*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

// plant with drag (Quadrotor3D.f_nominal, drag=True, payload=False; src/quad.py:256-381)
template <typename T>
__device__ inline void plant_eval(const DevModel<T>& m, const T* x, const T* u, T* f) {
  const T* q = x + 3; const T* v = x + 7; const T* r = x + 10;
  T R[9];
  rotmat(q, R);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  f[3] = T(0.5) * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = T(0.5) * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = T(0.5) * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = T(0.5) * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  const T aT = m.tmax * (u[0] + u[1] + u[2] + u[3]) / m.mass;
  T vb[3], ad[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T sg = T((vb[i] > 0) - (vb[i] < 0));
    ad[i] = -m.aero_drag * vb[i] * vb[i] * sg / m.mass - m.rotor_drag[i] * vb[i] / m.mass;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) f[7 + i] = R[3 * i] * ad[0] + R[3 * i + 1] * ad[1] + R[3 * i + 2] * (ad[2] + aT);
  f[9] -= m.g;
  T ty = 0, tx = 0, tz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ty += u[j] * m.yf[j]; tx += u[j] * m.xf[j]; tz += u[j] * m.zl[j]; }
  f[10] = (m.tmax * ty + (m.J[1] - m.J[2]) * r[1] * r[2]) / m.J[0];
  f[11] = (-m.tmax * tx + (m.J[2] - m.J[0]) * r[2] * r[0]) / m.J[1];
  f[12] = (m.tmax * tz + (m.J[0] - m.J[1]) * r[0] * r[1]) / m.J[2];
}
template <typename T>
__device__ inline void plant_rk4(const DevModel<T>& m, T* x, const T* uin, T dt) {
  T u[4], k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = tmin(T(1), tmax(T(0), uin[j]));
  plant_eval(m, x, u, k1);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k1[i];
  plant_eval(m, xt, u, k2);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k2[i];
  plant_eval(m, xt, u, k3);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
  plant_eval(m, xt, u, k4);
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

// ------------------------------------------------------------------ shooting
// pass 1: lane per interval, 4 RK substages; writes records + gap c_i = Phi_i - X_{i+1}
template <typename T>
__device__ inline void shoot_states(const DevModel<T>& m, T* S, const Lds& L, bool gp) {
  const int N = m.N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    T x[NX], u[NU], k[NX], xt[NX], acc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = S[L.X + i * NX + j];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = S[L.U + i * NU + j];
    const T* al = gp ? S + L.alpha : nullptr;
    T* sub = S + L.sub + i * 4 * SUBW;
    model_eval(m, x, u, al, S + L.basis, k, sub);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = k[j]; xt[j] = x[j] + m.h / 2 * k[j]; }
    model_eval(m, xt, u, al, S + L.basis, k, sub + SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] += 2 * k[j]; xt[j] = x[j] + m.h / 2 * k[j]; }
    model_eval(m, xt, u, al, S + L.basis, k, sub + 2 * SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] += 2 * k[j]; xt[j] = x[j] + m.h * k[j]; }
    model_eval(m, xt, u, al, S + L.basis, k, sub + 3 * SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const T phi = x[j] + m.h / 6 * (acc[j] + k[j]);
      S[L.c + i * NX + j] = phi - S[L.X + (i + 1) * NX + j];
    }
  }
}
// pass 2: item = (interval i, column j of [A|B], j = 3..16); columns 0..2 of A are [I;0] exactly
template <typename T>
__device__ inline void shoot_sens(const DevModel<T>& m, T* S, const Lds& L) {
  const int N = m.N;
  const T a_s[4] = {T(0), T(0.5), T(0.5), T(1)};
  const T w_s[4] = {T(1), T(2), T(2), T(1)};
  for (int it = threadIdx.x; it < N * 14; it += blockDim.x) {
    const int i = it / 14, j = 3 + it % 14;
    const bool ucol = j >= NX;
    T Sp[NX], acc[NX], Z[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) { Sp[r] = 0; acc[r] = 0; }
    // input-column constants of J_u
    T jur[3] = {0, 0, 0};
    if (ucol) {
      const int c = j - NX;
      jur[0] = m.tmax * m.yf[c] / m.J[0]; jur[1] = -m.tmax * m.xf[c] / m.J[1]; jur[2] = m.tmax * m.zl[c] / m.J[2];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T* sub = S + L.sub + (i * 4 + s) * SUBW;
      const T hs = m.h * a_s[s];
#pragma unroll
      for (int r = 0; r < NX; ++r) Z[r] = ((r == j) ? T(1) : T(0)) + hs * Sp[r];
      const T qw = sub[3], qx = sub[4], qy = sub[5], qz = sub[6];
      const T r0 = sub[10], r1 = sub[11], r2 = sub[12];
      T Sn[NX];
      Sn[0] = Z[7]; Sn[1] = Z[8]; Sn[2] = Z[9];
      Sn[3] = T(0.5) * (-r0 * Z[4] - r1 * Z[5] - r2 * Z[6] - qx * Z[10] - qy * Z[11] - qz * Z[12]);
      Sn[4] = T(0.5) * (r0 * Z[3] + r2 * Z[5] - r1 * Z[6] + qw * Z[10] - qz * Z[11] + qy * Z[12]);
      Sn[5] = T(0.5) * (r1 * Z[3] - r2 * Z[4] + r0 * Z[6] + qz * Z[10] + qw * Z[11] - qx * Z[12]);
      Sn[6] = T(0.5) * (r2 * Z[3] + r1 * Z[4] - r0 * Z[5] - qy * Z[10] + qx * Z[11] + qw * Z[12]);
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        T t = ucol ? sub[34 + row] * m.tmax / m.mass : T(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) t += sub[13 + row * 4 + c] * Z[3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) t += sub[25 + row * 3 + c] * Z[7 + c];
        Sn[7 + row] = t;
      }
      Sn[10] = jur[0] + (m.J[1] - m.J[2]) / m.J[0] * (r2 * Z[11] + r1 * Z[12]);
      Sn[11] = jur[1] + (m.J[2] - m.J[0]) / m.J[1] * (r2 * Z[10] + r0 * Z[12]);
      Sn[12] = jur[2] + (m.J[0] - m.J[1]) / m.J[2] * (r1 * Z[10] + r0 * Z[11]);
#pragma unroll
      for (int r = 0; r < NX; ++r) { acc[r] += w_s[s] * Sn[r]; Sp[r] = Sn[r]; }
    }
    T* AB = S + L.AB + i * NAB;
#pragma unroll
    for (int r = 0; r < NX; ++r) AB[r * NY + j] = ((r == j) ? T(1) : T(0)) + m.h / 6 * acc[r];
  }
  for (int it = threadIdx.x; it < N * NX * 3; it += blockDim.x) {
    const int i = it / (NX * 3), r = (it / 3) % NX, j = it % 3;
    S[L.AB + i * NAB + r * NY + j] = (r == j) ? T(1) : T(0);
  }
}

// ------------------------------------------------------------------ QP pieces
// Stage data of the QP in (dx, du):  Q_i = h*W_x (i<N) / W_e (i=N), R = h*W_u (diagonal),
// q_i = Q_i (X_i - xref_i), rho_i = R (U_i - uref_i); bounds lb = ulb - U, ub = uub - U.
template <typename T> __device__ inline T Qd(const DevModel<T>& m, int i, int k) { return i < m.N ? m.h * m.W[k] : m.We[k]; }
template <typename T> __device__ inline T xref(const T* S, const Lds& L, const DevModel<T>& m, int i, int k) {
  return i < m.N ? S[L.yref + i * NY + k] : S[L.yref + m.N * NY + k];
}

// forward rollout of the affine dynamics: dx_0 given, dx_{i+1} = A dx_i + B z_i + (with_c ? c_i : 0)
template <typename T>
__device__ inline void rollout(const DevModel<T>& m, T* S, const Lds& L, int dxo, int zo, bool with_c) {
  const int N = m.N;
  for (int i = 0; i < N; ++i) {
    if (threadIdx.x < NX) {
      const int r = threadIdx.x;
      const T* AB = S + L.AB + i * NAB + r * NY;
      T t = with_c ? S[L.c + i * NX + r] : T(0);
#pragma unroll
      for (int k = 0; k < NX; ++k) t += AB[k] * S[dxo + i * NX + k];
#pragma unroll
      for (int k = 0; k < NU; ++k) t += AB[NX + k] * S[zo + i * NU + k];
      S[dxo + (i + 1) * NX + r] = t;
    }
    __syncthreads();
  }
}

// adjoint sweep: grad = d/dz of the QP objective at (dx(z), z)
template <typename T>
__device__ inline void adjoint(const DevModel<T>& m, T* S, const Lds& L) {
  const int N = m.N;
  if (threadIdx.x < NX) {
    const int k = threadIdx.x;
    S[L.piv + k] = Qd(m, N, k) * (S[L.dx + N * NX + k] + S[L.X + N * NX + k] - xref(S, L, m, N, k));
  }
  __syncthreads();
  for (int i = N - 1; i >= 0; --i) {
    const int cur = ((N - 1 - i) & 1) * NX, nxt = NX - cur;  // piv ping-pong
    if (threadIdx.x < NY) {
      const int a = threadIdx.x;
      const T* AB = S + L.AB + i * NAB + a;
      T t = 0;
#pragma unroll
      for (int k = 0; k < NX; ++k) t += AB[k * NY] * S[L.piv + cur + k];
      if (a < NX) S[L.piv + nxt + a] = t + Qd(m, i, a) * (S[L.dx + i * NX + a] + S[L.X + i * NX + a] - xref(S, L, m, i, a));
      else {
        const int j = a - NX;
        S[L.grad + i * NU + j] = t + m.h * m.W[NX + j] * (S[L.z + i * NU + j] + S[L.U + i * NU + j] - S[L.yref + i * NY + NX + j]);
      }
    }
    __syncthreads();
  }
}

// Backward Riccati sweep.  with_matrix: recompute P_i, K_i, Lambda_i^-1 for R~ = R + ll/sl + lu/su.
// Always: vector recursion for the linear term rho (S[L.rho]) -> feed-forward kv.
// Returns false if a stage Hessian was not positive definite.
template <typename T>
__device__ inline bool riccati_backward(const DevModel<T>& m, T* S, const Lds& L, bool with_matrix, bool polish) {
  const int N = m.N, tid = threadIdx.x, NT = blockDim.x;
  bool ok = true;
  if (with_matrix)
    for (int it = tid; it < NX * NX; it += NT) S[L.P + it] = (it / NX == it % NX) ? m.We[it / NX] : T(0);
  if (tid < NX) S[L.pv + tid] = 0;
  __syncthreads();
  for (int i = N - 1; i >= 0; --i) {
    const int pc = ((N - 1 - i) & 1), pn = 1 - pc;
    const T* AB = S + L.AB + i * NAB;
    const T* Pn = S + L.P + pc * NX * NX;
    const T* pvn = S + L.pv + pc * NX;
    // phase A: T1 = P_{i+1} [A|B];  tv = [A|B]^T p_{i+1}
    if (with_matrix)
      for (int it = tid; it < NAB; it += NT) {
        const int r = it / NY, c = it % NY;
        T t = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) t += Pn[r * NX + k] * AB[k * NY + c];
        S[L.T1 + it] = t;
      }
    for (int a = tid; a < NY; a += NT) {
      T t = 0;
#pragma unroll
      for (int k = 0; k < NX; ++k) t += AB[k * NY + a] * pvn[k];
      S[L.tv + a] = t;
    }
    __syncthreads();
    // phase B: F = [A|B]^T T1 (lower triangle)
    if (with_matrix) {
      for (int it = tid; it < NY * (NY + 1) / 2; it += NT) {
        // unrank (a >= b) from it
        int a = 0, rem = it;
        while (rem > a) { rem -= a + 1; ++a; }
        const int b = rem;
        T t = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) t += AB[k * NY + a] * S[L.T1 + k * NY + b];
        S[L.F + a * NY + b] = t;
      }
      __syncthreads();
      // phase C: Lambda = R~ + F_uu ; explicit inverse of the 4x4 SPD block (one lane)
      if (tid == 0) {
        T Lm[4][4], Li[4][4];
        bool pd = true;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) Lm[a][b] = S[L.F + (NX + a) * NY + NX + b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int iv = i * NU + a;
          if (!polish) Lm[a][a] += m.h * m.W[NX + a] + S[L.ll + iv] / S[L.sl + iv] + S[L.lu + iv] / S[L.su + iv];
          else if (S[L.act + iv] == T(0)) Lm[a][a] += m.h * m.W[NX + a];
          else {  // input pinned on a bound: eliminate it from the stage Hessian
#pragma unroll
            for (int b = 0; b < 4; ++b) { if (b < a) Lm[a][b] = 0; if (b > a) Lm[b][a] = 0; }
            Lm[a][a] = 1;
          }
        }
        // Cholesky Lm = G G^T (in place, lower)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          T d = Lm[j][j];
#pragma unroll
          for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k];
          if (!(d > 0)) { pd = false; d = 1; }
          d = tsqrt(d);
          Lm[j][j] = d;
          const T id = 1 / d;
#pragma unroll
          for (int a = j + 1; a < 4; ++a) {
            T s = Lm[a][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[a][k] * Lm[j][k];
            Lm[a][j] = s * id;
          }
        }
        // G^-1 (lower), then Lambda^-1 = G^-T G^-1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Li[j][j] = 1 / Lm[j][j];
#pragma unroll
          for (int a = j + 1; a < 4; ++a) {
            T s = 0;
#pragma unroll
            for (int k = j; k < a; ++k) s -= Lm[a][k] * Li[k][j];
            Li[a][j] = s / Lm[a][a];
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            T s = 0;
#pragma unroll
            for (int k = a; k < 4; ++k) s += Li[k][a] * Li[k][b];
            S[L.Linv + i * 16 + a * 4 + b] = s;
            S[L.Linv + i * 16 + b * 4 + a] = s;
          }
        S[L.red + 8] = pd ? T(0) : T(1);
      }
      __syncthreads();
      if (S[L.red + 8] != T(0)) ok = false;
      // phase D: K = -Lambda^-1 M,  M = F_ux (4x13)
      for (int it = tid; it < NU * NX; it += NT) {
        const int j = it / NX, b = it % NX;
        T t = 0;
#pragma unroll
        for (int k = 0; k < NU; ++k) t -= S[L.Linv + i * 16 + j * 4 + k] * S[L.F + (NX + k) * NY + b];
        if (polish && S[L.act + i * NU + j] != T(0)) t = 0;
        S[L.K + i * NU * NX + it] = t;
      }
    }
    // vector: gt = rho_i + B^T p_{i+1}
    if (tid < NU) S[L.tv + NY + tid] = (polish && S[L.act + i * NU + tid] != T(0)) ? T(0) : S[L.rho + i * NU + tid] + S[L.tv + NX + tid];
    __syncthreads();
    // phase E: P_i = Q + F_xx + M^T K ; k_i = -Lambda^-1 gt ; p_i = A^T p_{i+1} + K^T gt
    if (with_matrix && i > 0) {
      T* Po = S + L.P + pn * NX * NX;
      for (int it = tid; it < NX * (NX + 1) / 2; it += NT) {
        int a = 0, rem = it;
        while (rem > a) { rem -= a + 1; ++a; }
        const int b = rem;
        T t = S[L.F + a * NY + b] + (a == b ? m.h * m.W[a] : T(0));
#pragma unroll
        for (int k = 0; k < NU; ++k) t += S[L.F + (NX + k) * NY + a] * S[L.K + i * NU * NX + k * NX + b];
        Po[a * NX + b] = t;
        Po[b * NX + a] = t;
      }
    }
    if (tid < NX) {
      T t = S[L.tv + tid];
#pragma unroll
      for (int k = 0; k < NU; ++k) t += S[L.K + i * NU * NX + k * NX + tid] * S[L.tv + NY + k];
      S[L.pv + pn * NX + tid] = t;
    } else if (tid >= 16 && tid < 16 + NU) {
      const int j = tid - 16;
      T t = 0;
#pragma unroll
      for (int k = 0; k < NU; ++k) t -= S[L.Linv + i * 16 + j * 4 + k] * S[L.tv + NY + k];
      S[L.kv + i * NU + j] = (polish && S[L.act + i * NU + j] != T(0)) ? T(0) : t;
    }
    __syncthreads();
  }
  return ok;
}

// forward sweep: Dx_0 = 0; dz_i = K_i Dx_i + k_i ; Dx_{i+1} = A Dx_i + B dz_i   (out: S[dzo], S[L.Dx])
template <typename T>
__device__ inline void riccati_forward(const DevModel<T>& m, T* S, const Lds& L, int dzo) {
  const int N = m.N, tid = threadIdx.x;
  if (tid < NX) S[L.Dx + tid] = 0;
  __syncthreads();
  for (int i = 0; i < N; ++i) {
    if (tid < NU) {
      T t = S[L.kv + i * NU + tid];
#pragma unroll
      for (int k = 0; k < NX; ++k) t += S[L.K + i * NU * NX + tid * NX + k] * S[L.Dx + i * NX + k];
      S[dzo + i * NU + tid] = t;
    }
    __syncthreads();
    if (tid < NX) {
      const T* AB = S + L.AB + i * NAB + tid * NY;
      T t = 0;
#pragma unroll
      for (int k = 0; k < NX; ++k) t += AB[k] * S[L.Dx + i * NX + k];
#pragma unroll
      for (int k = 0; k < NU; ++k) t += AB[NX + k] * S[dzo + i * NU + k];
      S[L.Dx + (i + 1) * NX + tid] = t;
    }
    __syncthreads();
  }
}

// Mehrotra predictor-corrector iterations on  min 1/2 z'Hz + g'z, lb <= z <= ub  with H, g implicit
// in the stage data; every Newton system is one Riccati factorisation + two vector sweeps.
// Continues from the current (z, sl, su, ll, lu, dx, grad) until  |r_d| <= tol*gm and mu <= tol.
// returns 0 converged / 1 NaN / 2 iteration cap / 4 stage Hessian not positive definite
template <typename T>
__device__ inline int ipm_run(const DevModel<T>& m, T* S, const Lds& L, const T tol, const T gm, int& it) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x, NT = blockDim.x;
  T* red = S + L.red;
  int status = 2;
  const int maxit = m.qp_max_iter;
  for (; it < maxit; ++it) {
    // residuals at the current point (grad is current)
    T rdm = 0, mu = 0;
    for (int i = tid; i < nv; i += NT) {
      rdm = tmax(rdm, tabs(S[L.grad + i] - S[L.ll + i] + S[L.lu + i]));
      mu += S[L.sl + i] * S[L.ll + i] + S[L.su + i] * S[L.lu + i];
    }
    rdm = block_reduce<T, 1>(rdm, red);
    mu = block_reduce<T, 0>(mu, red) / (2 * nv);
    if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
    if (rdm <= tol * gm && mu <= tol) { status = 0; break; }
    // predictor: (H + Sigma) dza = -grad
    for (int i = tid; i < nv; i += NT) S[L.rho + i] = S[L.grad + i];
    __syncthreads();
    if (!riccati_backward(m, S, L, true, false)) { status = 4; break; }
    riccati_forward(m, S, L, L.dza);
    T aff = 1;
    for (int i = tid; i < nv; i += NT) {
      const T d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const T dl = -ll - ll / sl * d, du = -lu + lu / su * d;
      if (d < 0) aff = tmin(aff, -sl / d);
      if (d > 0) aff = tmin(aff, su / d);
      if (dl < 0) aff = tmin(aff, -ll / dl);
      if (du < 0) aff = tmin(aff, -lu / du);
    }
    aff = block_reduce<T, 2>(aff, red);
    T mua = 0;
    for (int i = tid; i < nv; i += NT) {
      const T d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const T dl = -ll - ll / sl * d, du = -lu + lu / su * d;
      mua += (sl + aff * d) * (ll + aff * dl) + (su - aff * d) * (lu + aff * du);
    }
    mua = block_reduce<T, 0>(mua, red) / (2 * nv);
    T sigma = mua / mu;
    sigma = sigma * sigma * sigma;
    // corrector rhs r = -rd + rcl/sl - rcu/su ; linear term rho = -r
    for (int i = tid; i < nv; i += NT) {
      const T d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const T dl = -ll - ll / sl * d, du = -lu + lu / su * d;
      const T rcl = -sl * ll + sigma * mu - d * dl;
      const T rcu = -su * lu + sigma * mu + d * du;
      const T rd = S[L.grad + i] - ll + lu;
      S[L.rho + i] = rd - rcl / sl + rcu / su;
    }
    __syncthreads();
    riccati_backward(m, S, L, false, false);
    riccati_forward(m, S, L, L.dz);
    T ap = 1, ad = 1;
    for (int i = tid; i < nv; i += NT) {
      const T da = S[L.dza + i], d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const T dla = -ll - ll / sl * da, dua = -lu + lu / su * da;
      const T rcl = -sl * ll + sigma * mu - da * dla;
      const T rcu = -su * lu + sigma * mu + da * dua;
      const T dl = (rcl - ll * d) / sl, du = (rcu + lu * d) / su;
      if (d < 0) ap = tmin(ap, -sl / d);
      if (d > 0) ap = tmin(ap, su / d);
      if (dl < 0) ad = tmin(ad, -ll / dl);
      if (du < 0) ad = tmin(ad, -lu / du);
    }
    ap = block_reduce<T, 2>(ap, red);
    ad = block_reduce<T, 2>(ad, red);
    const T tau = tmax(T(0.995), 1 - mu);
    ap = tmin(T(1), tau * ap);
    ad = tmin(T(1), tau * ad);
    for (int i = tid; i < nv; i += NT) {
      const T da = S[L.dza + i], d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const T dla = -ll - ll / sl * da, dua = -lu + lu / su * da;
      const T rcl = -sl * ll + sigma * mu - da * dla;
      const T rcu = -su * lu + sigma * mu + da * dua;
      const T dl = (rcl - ll * d) / sl, du = (rcu + lu * d) / su;
      S[L.z + i] += ap * d; S[L.sl + i] = sl + ap * d; S[L.su + i] = su - ap * d;
      S[L.ll + i] = ll + ad * dl; S[L.lu + i] = lu + ad * du;
    }
    for (int i = tid; i < (N + 1) * NX; i += NT) S[L.dx + i] += ap * S[L.Dx + i];
    __syncthreads();
    adjoint(m, S, L);
  }
  return status;
}

// Active-set polish: starting from the IPM point, pin the inputs the IPM identifies as active,
// and take Newton steps on the free set (masked Riccati) with a ratio test, releasing inputs whose
// multiplier has the wrong sign and pinning inputs that block.  Ends on an exact KKT point of the
// QP (to rounding), which an interior method only approaches like sqrt(mu) on weakly active bounds.
template <typename T>
__device__ inline bool polish(const DevModel<T>& m, T* S, const Lds& L, const T gm, int& passes) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x, NT = blockDim.x;
  T* red = S + L.red;
  for (int i = tid; i < nv; i += NT)
    S[L.act + i] = S[L.ll + i] > S[L.sl + i] ? T(-1) : (S[L.lu + i] > S[L.su + i] ? T(1) : T(0));
  __syncthreads();
  const T tolm = 64 * m.eps * gm;  // multiplier sign / stationarity
  const T tolb = 16 * m.eps;       // bound proximity (bounds are O(1))
  bool refactor = true, settled = false, full = false;
  T gF_prev = T(1e30);
  for (passes = 0; passes < m.polish_max; ++passes) {
    for (int i = tid; i < nv; i += NT) {
      const T a = S[L.act + i];
      if (a < 0) S[L.z + i] = m.ulb[i & 3] - S[L.U + i];
      else if (a > 0) S[L.z + i] = m.uub[i & 3] - S[L.U + i];
    }
    if (tid < NX) S[L.dx + tid] = S[L.x0 + tid] - S[L.X + tid];
    __syncthreads();
    rollout(m, S, L, L.dx, L.z, true);
    adjoint(m, S, L);
    // stationarity on the free set, worst multiplier sign violation on the pinned set
    T gF = 0, vmax = 0;
    for (int i = tid; i < nv; i += NT) {
      const T a = S[L.act + i], g = S[L.grad + i];
      if (a == T(0)) gF = tmax(gF, tabs(g));
      else vmax = tmax(vmax, a < 0 ? -g : g);
    }
    gF = block_reduce<T, 1>(gF, red);
    vmax = block_reduce<T, 1>(vmax, red);
    if (!(gF == gF)) return false;
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) { int na = 0; for (int i = 0; i < nv; ++i) na += S[L.act + i] != T(0); printf("  polish pass %d full %d gF %.3e vmax %.3e tolm %.3e nact %d\n", passes, (int)full, (double)gF, (double)vmax, (double)tolm, na); }
#endif
    if (full) {
      // the point minimises the QP on the working set: multipliers are meaningful here only
      if (vmax > tolm) {
        for (int i = tid; i < nv; i += NT) {
          const T a = S[L.act + i], g = S[L.grad + i];
          if (a != T(0) && (a < 0 ? -g : g) >= vmax) S[L.act + i] = 0;  // release the worst one
        }
        refactor = true;
        gF_prev = T(1e30);
        __syncthreads();
      } else if (gF <= tolm || gF > T(0.25) * gF_prev) {
        settled = true;  // stationary, or refinement stagnated at the rounding level
        break;
      } else {
        gF_prev = gF;
      }
    }
    for (int i = tid; i < nv; i += NT) S[L.rho + i] = S[L.grad + i];
    __syncthreads();
    if (!riccati_backward(m, S, L, refactor, true)) return false;
    refactor = false;
    riccati_forward(m, S, L, L.dz);
    T alpha = 1;
    for (int i = tid; i < nv; i += NT) {
      if (S[L.act + i] != T(0)) continue;
      const T d = S[L.dz + i], z = S[L.z + i];
      const T lb = m.ulb[i & 3] - S[L.U + i], ub = m.uub[i & 3] - S[L.U + i];
      if (d < 0) alpha = tmin(alpha, tmax(T(0), (lb - z) / d));
      if (d > 0) alpha = tmin(alpha, tmax(T(0), (ub - z) / d));
    }
    alpha = block_reduce<T, 2>(alpha, red);
    T nblk = 0;
    for (int i = tid; i < nv; i += NT) {
      if (S[L.act + i] != T(0)) continue;
      const T d = S[L.dz + i], z = S[L.z + i] + alpha * d;
      const T lb = m.ulb[i & 3] - S[L.U + i], ub = m.uub[i & 3] - S[L.U + i];
      S[L.z + i] = z;
      if (alpha < T(1)) {
        if (d < 0 && z <= lb + tolb) { S[L.act + i] = -1; nblk += 1; }
        else if (d > 0 && z >= ub - tolb) { S[L.act + i] = 1; nblk += 1; }
      }
    }
    nblk = block_reduce<T, 0>(nblk, red);
    full = nblk == T(0);
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("     alpha %.6e nblk %g\n", (double)alpha, (double)nblk);
#endif
    if (nblk > 0) refactor = true;
    __syncthreads();
  }
  return settled;
}

// Box-QP solve: IPM to the hand-over tolerance, then active-set polish; if the polish does not
// settle (degenerate cycling), fall back to IPM iterations down to the final tolerance.
// On exit S[L.z] holds the solution; returns IPM iterations + polish passes.
template <typename T>
__device__ inline int solve_qp(const DevModel<T>& m, T* S, const Lds& L, int* status) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x, NT = blockDim.x;
  T* red = S + L.red;
  // interior start
  for (int i = tid; i < nv; i += NT) {
    const int j = i & 3;
    const T lb = m.ulb[j] - S[L.U + i], ub = m.uub[j] - S[L.U + i], w = ub - lb;
    const T z0 = tmin(tmax(T(0), lb + T(0.1) * w), ub - T(0.1) * w);
    S[L.z + i] = z0; S[L.sl + i] = z0 - lb; S[L.su + i] = ub - z0;
  }
  if (tid < NX) S[L.dx + tid] = S[L.x0 + tid] - S[L.X + tid];
  __syncthreads();
  rollout(m, S, L, L.dx, L.z, true);
  adjoint(m, S, L);
  T gm = 1;
  for (int i = tid; i < nv; i += NT) gm = tmax(gm, tabs(S[L.grad + i]));
  gm = block_reduce<T, 1>(gm, red);
  for (int i = tid; i < nv; i += NT) { S[L.ll + i] = T(0.1) * gm / S[L.sl + i]; S[L.lu + i] = T(0.1) * gm / S[L.su + i]; }
  __syncthreads();
  int it = 0, passes = 0;
  int st = ipm_run(m, S, L, m.polish_max > 0 ? m.ipm_tol : m.qp_tol, gm, it);
  if (st == 0 && m.polish_max > 0) {
    for (int i = tid; i < nv; i += NT) S[L.dza + i] = S[L.z + i];
    __syncthreads();
    if (!polish(m, S, L, gm, passes)) {
      for (int i = tid; i < nv; i += NT) S[L.z + i] = S[L.dza + i];
      if (tid < NX) S[L.dx + tid] = S[L.x0 + tid] - S[L.X + tid];
      __syncthreads();
      rollout(m, S, L, L.dx, L.z, true);
      adjoint(m, S, L);
      st = ipm_run(m, S, L, m.qp_tol, gm, it);
    }
  }
  *status = st;
  return it + passes;
}

// ------------------------------------------------------------------ RGP regress (3 axes, one new point each)
// RGP.regress / RGP.predict of the reference (src/gp/RGP.py:199-208,303-330), scalar new point:
//   J = k* Kx^-1 ; mu_p = J mu ; Cp = sf2 - J k* + J C J^T ; G = C J^T/(Cp + sn2) ;
//   mu += G (y - mu_p) ; C -= G (J C)      (not symmetrised, as in the reference)
template <typename T>
__device__ inline void rgp_regress(const DevModel<T>& m, T* S, const Lds& L, T* gmu, T* gC, const T* vb, const T* ad) {
  const int n = m.nb, tid = threadIdx.x, NT = blockDim.x, n3 = 3 * n, nn = n * n;
  T* C = S + L.rgp;
  T* ks = C + al4(3 * nn);
  T* Jt = ks + al4(n3);
  T* JC = Jt + al4(n3);
  T* CJ = JC + al4(n3);
  T* mu = CJ + al4(n3);
  T* sc = mu + al4(n3);  // per axis: [mu_p, Jk, JCJ]
  for (int i = tid; i < 3 * nn; i += NT) C[i] = gC[i];
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n;
    const T dl = vb[d] - m.basis[i];
    ks[i] = m.sf2[d] * texp(T(-0.5) * dl * dl * m.L2inv[d]);
    mu[i] = gmu[i];
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    T t = 0;
    for (int k = 0; k < n; ++k) t += ks[d * n + k] * m.Kxinv[d * nn + k * n + j];
    Jt[i] = t;
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    T t = 0, s = 0;
    for (int k = 0; k < n; ++k) { t += Jt[d * n + k] * C[d * nn + k * n + j]; s += C[d * nn + j * n + k] * Jt[d * n + k]; }
    JC[i] = t;
    CJ[i] = s;
  }
  __syncthreads();
  if (tid < 3) {
    const int d = tid;
    T mup = 0, Jk = 0, JCJ = 0;
    for (int k = 0; k < n; ++k) { mup += Jt[d * n + k] * mu[d * n + k]; Jk += Jt[d * n + k] * ks[d * n + k]; JCJ += JC[d * n + k] * Jt[d * n + k]; }
    const T Cp = m.sf2[d] - Jk + JCJ;
    sc[d * 4] = (ad[d] - mup);
    sc[d * 4 + 1] = 1 / (Cp + m.sn2[d]);
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n;
    gmu[i] = mu[i] + CJ[i] * sc[d * 4 + 1] * sc[d * 4];
  }
  for (int i = tid; i < 3 * nn; i += NT) {
    const int d = i / nn, r = (i / n) % n, c = i % n;
    gC[i] = C[i] - CJ[d * n + r] * sc[d * 4 + 1] * JC[d * n + c];
  }
}

// ------------------------------------------------------------------ the fused step kernel
template <typename T>
__global__ void __launch_bounds__(256) step_kernel(const DevModel<T> m, const DevState<T> st, const int mode) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* S = reinterpret_cast<T*>(smem_raw);
  const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int N = m.N, nb = m.nb;
  const Lds L = lds_layout(N, nb);
  const bool gp = nb > 0;
  // ---- load persistent state (lane-contiguous records)
  T* gX = st.X + (size_t)b * (N + 1) * NX;
  T* gU = st.U + (size_t)b * N * NU;
  for (int i = tid; i < (N + 1) * NX; i += NT) S[L.X + i] = gX[i];
  for (int i = tid; i < N * NU; i += NT) S[L.U + i] = gU[i];
  if (tid < NX) S[L.x0 + tid] = st.x_meas[(size_t)b * NX + tid];
  const int idx = st.idx[b];
  if (mode & MODE_TRAJ) {
    // get_reference_chunk (src/utils/utils.py:897-931) + set_reference_trajectory (src/quad_opt.py:295-317)
    const int len = st.tlen[b], skip = m.skip;
    const long left = (long)len - idx;
    int have = 0;
    if (left > (long)N * skip) have = N;
    else if (left > skip - 1) { have = (int)((left + skip - 1) / skip); if (have > N) have = N; }
    const T* tr = st.traj + (size_t)b * m.Tmax * NX;
    for (int it = tid; it < N * NY; it += NT) {
      const int j = it / NY, k = it % NY;
      const long row = j < have ? (long)idx + (long)j * skip : (long)len - 1;
      const T v = k < NX ? tr[row * NX + k] : m.uref[k - NX];
      S[L.yref + it] = v;
      st.yref[(size_t)b * N * NY + it] = v;
    }
    if (tid < NX) {
      const long row = (N - 1) < have ? (long)idx + (long)(N - 1) * skip : (long)len - 1;
      const T v = tr[row * NX + tid];
      S[L.yref + N * NY + tid] = v;
      st.yrefN[(size_t)b * NX + tid] = v;
    }
  } else {
    for (int it = tid; it < N * NY; it += NT) S[L.yref + it] = st.yref[(size_t)b * N * NY + it];
    if (tid < NX) S[L.yref + N * NY + tid] = st.yrefN[(size_t)b * NX + tid];
  }
  T* gmu = st.mu + (size_t)b * 3 * nb;
  if (gp) {
    // alpha = Kx^-1 mu  (the OCP model evaluates k*(v_b) Kx^-1 p, src/gp/RGP.py:250-254)
    for (int i = tid; i < 3 * nb; i += NT) {
      const int d = i / nb, r = i % nb;
      T t = 0;
      for (int k = 0; k < nb; ++k) t += m.Kxinv[d * nb * nb + r * nb + k] * gmu[d * nb + k];
      S[L.alpha + i] = t;
      S[L.basis + i] = m.basis[i];
    }
  }
  __syncthreads();
  // ---- 1. shooting
  shoot_states(m, S, L, gp);
  __syncthreads();
  shoot_sens(m, S, L);
  __syncthreads();
  // ---- 2. QP
  int status = 0;
  const int iters = solve_qp(m, S, L, &status);
  // ---- 3. expand with a fresh rollout from the converged z, full step
  if (tid < NX) S[L.dx + tid] = S[L.x0 + tid] - S[L.X + tid];
  __syncthreads();
  rollout(m, S, L, L.dx, L.z, true);
  for (int i = tid; i < (N + 1) * NX; i += NT) { const T v = S[L.X + i] + S[L.dx + i]; S[L.X + i] = v; gX[i] = v; }
  for (int i = tid; i < N * NU; i += NT) { const T v = S[L.U + i] + S[L.z + i]; S[L.U + i] = v; gU[i] = v; }
  __syncthreads();
  // cost at the new iterate
  T cst = 0;
  bool bad = false;
  for (int it = tid; it < N * NY; it += NT) {
    const int i = it / NY, k = it % NY;
    const T v = k < NX ? S[L.X + i * NX + k] : S[L.U + i * NU + k - NX];
    const T e = v - S[L.yref + it];
    cst += T(0.5) * m.h * m.W[k] * e * e;
    if (!(v == v)) bad = true;
  }
  if (tid < NX) { const T e = S[L.X + N * NX + tid] - S[L.yref + N * NY + tid]; cst += T(0.5) * m.We[tid] * e * e; }
  cst = block_reduce<T, 0>(cst, S + L.red);
  const T badf = block_reduce<T, 1>(bad ? T(1) : T(0), S + L.red);
  if (badf != T(0)) status = 1;
  if (tid == 0) { st.cost[b] = cst; st.status[b] = status; st.qp_iter[b] = iters; }
  if (tid < NU) st.w[(size_t)b * NU + tid] = S[L.U + tid];
  if (!(mode & MODE_POST)) return;
  // ---- 4. post: nominal prediction, cursor, drag estimate, RGP regress, statistics
  __syncthreads();
  T* sc = S + L.rgp + L.total - L.rgp - 64;  // tail of the union (not touched by rgp_regress)
  (void)sc;
  T* vbad = S + L.red + 16;  // [vb(3), ad(3)]
  if (tid == 0) {
    T x[NX], u[NU], xp[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = S[L.x0 + k];
#pragma unroll
    for (int k = 0; k < NU; ++k) u[k] = S[L.U + k];
    rk4_step(m, x, u, (const T*)nullptr, (const T*)nullptr, m.dt_pred, xp);
    // compute_a_drag (src/utils/utils.py:934-950) against the previous step's prediction
    T xq[NX];
    const bool hp = st.has_prev[b] != 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) xq[k] = hp ? st.xpp[(size_t)b * NX + k] : x[k];
    T R[9], Rq[9];
    rotmat(x + 3, R);
    rotmat(xq + 3, Rq);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const T vb = R[i] * x[7] + R[3 + i] * x[8] + R[6 + i] * x[9];
      const T vp = Rq[i] * xq[7] + Rq[3 + i] * xq[8] + Rq[6 + i] * xq[9];
      vbad[i] = vb;
      vbad[3 + i] = (vb - vp) / m.dt_pred;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) { st.xpred[(size_t)b * NX + k] = xp[k]; st.xpp[(size_t)b * NX + k] = xp[k]; }
    st.has_prev[b] = 1;
    st.idx[b] = idx + 1;
    // tracking statistic against the first row of the reference chunk
    T ep = 0, ev = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T a = x[k] - S[L.yref + k], c = x[7 + k] - S[L.yref + 7 + k];
      ep += a * a; ev += c * c;
    }
    T* gs = st.stats + (size_t)b * 4;
    gs[0] += ep; gs[1] += ev; gs[2] += 1; gs[3] = tmax(gs[3], ep);
  }
  __syncthreads();
  if (gp) rgp_regress(m, S, L, gmu, st.C + (size_t)b * 3 * nb * nb, vbad, vbad + 3);
}

// ------------------------------------------------------------------ small explicit-path kernels
template <typename T>
__global__ void predict_kernel(const DevModel<T> m, const T* x, const T* u, T dt, T* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  T xi[NX], ui[NU], xo[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k) xi[k] = x[(size_t)b * NX + k];
#pragma unroll
  for (int k = 0; k < NU; ++k) ui[k] = u[(size_t)b * NU + k];
  rk4_step(m, xi, ui, (const T*)nullptr, (const T*)nullptr, dt, xo);
#pragma unroll
  for (int k = 0; k < NX; ++k) out[(size_t)b * NX + k] = xo[k];
}

template <typename T>
__global__ void regress_kernel(const DevModel<T> m, const DevState<T> st, const T* vb, const T* ad) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* S = reinterpret_cast<T*>(smem_raw);
  const Lds L = lds_layout(m.N, m.nb);
  const int b = blockIdx.x;
  rgp_regress(m, S, L, st.mu + (size_t)b * 3 * m.nb, st.C + (size_t)b * 3 * m.nb * m.nb, vb + (size_t)b * 3, ad + (size_t)b * 3);
}

// closed-loop plant: n_sub RK4 substeps of the drag plant from the engine's plant state with input w
template <typename T>
__global__ void plant_kernel(const DevModel<T> m, T* xs, const T* w, int n_sub, T sim_dt, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  T x[NX], u[NU];
#pragma unroll
  for (int k = 0; k < NX; ++k) x[k] = xs[(size_t)b * NX + k];
#pragma unroll
  for (int k = 0; k < NU; ++k) u[k] = w[(size_t)b * NU + k];
  for (int s = 0; s < n_sub; ++s) plant_rk4(m, x, u, sim_dt);
#pragma unroll
  for (int k = 0; k < NX; ++k) xs[(size_t)b * NX + k] = x[k];
}

// reduce per-instance statistics to 5 numbers (sum, sum, sum, max, #failed)
template <typename T>
__global__ void stats_kernel(const T* stats, const int* status, int B, double* out5) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double (*sh)[256] = reinterpret_cast<double (*)[256]>(smem_raw);  // [5][256]
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    a0 += (double)stats[(size_t)b * 4]; a1 += (double)stats[(size_t)b * 4 + 1]; a2 += (double)stats[(size_t)b * 4 + 2];
    const double mx = (double)stats[(size_t)b * 4 + 3];
    a3 = a3 > mx ? a3 : mx;
    a4 += status[b] != 0 ? 1.0 : 0.0;
  }
  sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2; sh[3][threadIdx.x] = a3; sh[4][threadIdx.x] = a4;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < (int)blockDim.x; ++t) {
      sh[0][0] += sh[0][t]; sh[1][0] += sh[1][t]; sh[2][0] += sh[2][t]; sh[4][0] += sh[4][t];
      sh[3][0] = sh[3][0] > sh[3][t] ? sh[3][0] : sh[3][t];
    }
    for (int k = 0; k < 5; ++k) out5[k] = sh[k][0];
  }
}

}  // namespace mpcq
