// mpcq_kernels.hpp — device code of the batched MPC+RGP control step (gfx950 / CDNA4).
//
// One quadrotor per 64-lane wavefront (one workgroup = one wave).  All per-instance working
// data (iterate, shooting sensitivities, Riccati gains, IPM vectors, RGP covariance) is staged in
// LDS; HBM is touched once per step to load and once to store the persistent state, with
// lane-contiguous (coalesced) records.  Short vectors of the Riccati / adjoint / rollout
// recursions live in registers (lane a holds component a) and are broadcast with lane reads, so the
// N-stage sweeps run without LDS round trips or barriers on their critical path.
//
// Precision: the SQP iterate (X, U), the measurement, the reference and every difference that
// defines the QP data (x0 - X0, X_i - xref_i, U_i - uref_i, bounds, shooting gaps) are formed in
// double.  TQ (float or double) is the arithmetic of the model evaluation, the sensitivities and
// the QP solve.  TQ = double reproduces the fp64 oracle to ~1e-10; TQ = float is the fast path.
//
// Algorithm (same mathematical step as the reference's acados SQP-RTI call, restated in
// SURVEY App. A; not a translation of acados/HPIPM code):
//   1. multiple shooting: explicit RK4 (1 step) with forward sensitivities per interval
//   2. box-QP in du: Mehrotra predictor-corrector IPM to a hand-over tolerance, then an
//      active-set polish (Newton on the free set + ratio test) to an exact KKT point; every
//      Newton system is one Riccati factorisation of the stage-sparse problem
//      (O(N) memory, well conditioned; the condensed Hessian is never formed)
//   3. full step, cost, nominal RK4 prediction, body-frame drag estimate, 3 scalar RGP updates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcq {

constexpr int NX = 13, NU = 4, NY = 17;
constexpr int ABW = 16;          // row stride of AB' = columns 3..16 of [A|B] (cols 0..2 of A are [I;0])
constexpr int ABS = NX * ABW;    // per-stage stride of AB'
constexpr int KS = NU * ABW;     // per-stage stride of K (4 rows of 13, padded to 16)
constexpr int SUBW = 41;         // per (stage, RK substage) record: x_s(13) Jvq(12) Jvv(9) Rz(3) pad
constexpr int SUBS = 4 * SUBW + 1;  // per-stage stride of the records (odd: lanes of different stages hit different banks)

enum : int { MODE_TRAJ = 1, MODE_POST = 2 };

// Diagnostic build only (-DMPCQ_PROFILE, libmpcq_prof.so): per-phase shader-cycle totals per instance.
enum : int { PF_LOAD = 0, PF_SHOOT_X, PF_SHOOT_S, PF_FACTOR, PF_FWD, PF_BWD, PF_ADJ, PF_ROLL, PF_ELEM, PF_POST, PF_TOTAL, PF_N = 16 };
#ifdef MPCQ_PROFILE
struct Prof { unsigned long long acc[PF_N]; unsigned long long t; };
__device__ inline void pf_start(Prof& p) { p.t = __builtin_readcyclecounter(); }
__device__ inline void pf_stop(Prof& p, int k) { const unsigned long long n = __builtin_readcyclecounter(); p.acc[k] += n - p.t; p.t = n; }
#define PF_ARG , Prof& pf
#define PF_PASS , pf
#define PF_START() pf_start(pf)
#define PF_STOP(k) pf_stop(pf, k)
#else
#define PF_ARG
#define PF_PASS
#define PF_START()
#define PF_STOP(k)
#endif

template <typename TQ>
struct DevModel {
  int N, nb, skip, Tmax, B, qp_max_iter, polish_max;
  double h, dt_pred;
  double mass, J[3], tmax, xf[4], yf[4], zl[4], g;
  double W[NY], We[NX], ulb[NU], uub[NU], uref[NU];
  double rotor_drag[3], aero_drag;
  TQ qp_tol;    // final KKT tolerance (IPM-only fallback)
  TQ ipm_tol;   // IPM -> active-set polish hand-over tolerance
  TQ eps;       // unit roundoff scale of TQ used for KKT sign / bound tests
  TQ L2inv[3], sf2[3], sn2[3];
  const TQ* basis;  // [3*nb]
  const TQ* Kxinv;  // [3*nb*nb]
};

template <typename TQ>
struct DevState {
  double* X;        // [B][(N+1)*13]
  double* U;        // [B][N*4]
  TQ* mu;           // [B][3*nb]
  TQ* C;            // [B][3*nb*nb]
  double* xpp;      // [B][13]   x_pred of the previous step
  double* yref;     // [B][N*17]
  double* yrefN;    // [B][13]
  const double* traj;    // [B][Tmax][13]
  const double* x_meas;  // [B][13]
  double* w;        // [B][4]
  double* xpred;    // [B][13]
  double* cost;     // [B]
  double* stats;    // [B][4]
  int* has_prev;
  int* idx;
  const int* tlen;
  int* status;
  int* qp_iter;
  unsigned long long* prof;   // [B][PF_N] (diagnostic build only)
};

// ------------------------------------------------------------------ LDS layout
// doubles first (offsets in doubles from the LDS base), then the TQ region (offsets in TQ elements
// from the TQ base = base + dbytes).
struct Lds {
  int X, U, x0, dbytes;
  int AB, c, qv, r0, lb, ub, alpha, basis;
  int z, sl, su, ll, lu, grad, dza, dz, rho, kv, act, rt, dx, Dx, K, Linv, P, T1, F;
  int sub, rgp, qtotal;
};
__host__ __device__ inline int al4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline Lds lds_layout(int N, int nb) {
  Lds L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += al4(n); return r; };
  L.X = take((N + 1) * NX);
  L.U = take(N * NU);
  L.x0 = take(NX + 8);   // + [v_body(3), a_drag(3)] scratch of the post phase
  L.dbytes = o * 8;
  o = 0;
  const int nv = N * NU;
  L.AB = take(N * ABS);
  L.c = take(N * NX);
  L.qv = take((N + 1) * NX);
  L.r0 = take(nv); L.lb = take(nv); L.ub = take(nv);
  L.alpha = take(3 * nb);
  L.basis = take(3 * nb);
  const int u0 = o;  // ---- union: shooting records | QP workspace | RGP workspace
  L.sub = u0;
  const int sub_end = u0 + al4(N * SUBS);
  L.z = take(nv); L.sl = take(nv); L.su = take(nv); L.ll = take(nv); L.lu = take(nv);
  L.grad = take(nv); L.dza = take(nv); L.dz = take(nv); L.rho = take(nv); L.kv = take(nv); L.act = take(nv); L.rt = take(nv);
  L.dx = take((N + 1) * NX);
  L.Dx = take((N + 1) * NX);
  L.K = take(N * KS);
  L.Linv = take(N * 16);
  L.P = take(2 * ABS);
  L.T1 = take(ABS);
  L.F = take(14 * ABW);
  const int qp_end = o;
  L.rgp = u0;
  const int rgp_end = u0 + al4(3 * nb * nb) + 5 * al4(3 * nb) + 32;
  o = sub_end > qp_end ? sub_end : qp_end;
  if (rgp_end > o) o = rgp_end;
  L.qtotal = o;
  return L;
}
template <typename TQ> __host__ __device__ inline size_t lds_bytes(const Lds& L) { return (size_t)L.dbytes + (size_t)L.qtotal * sizeof(TQ); }

// ------------------------------------------------------------------ small helpers
template <typename T> struct alignas(16) V4 { T a, b, c, d; };

template <typename T> __device__ inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ inline T tmax(T a, T b) { return a > b ? a : b; }

// lane broadcast: `lane` must be wave-uniform (a constant after unrolling) -> v_readlane_b32 into an SGPR
__device__ inline int bc(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ inline float bc(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline double bc(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// DPP lane permutes inside 16-lane rows (quad_perm / row_ror), no LDS involved
template <int CTRL> __device__ inline int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ inline float dpp(float v) { return __int_as_float(dpp<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ inline double dpp(double v) {
  return __hiloint2double(dpp<CTRL>(__double2hiint(v)), dpp<CTRL>(__double2loint(v)));
}
// wave-wide reductions: butterfly inside each row of 16 with DPP, then the 4 row results through SGPRs
template <typename T, typename OP> __device__ inline T wave_reduce(T v, OP op) {
  v = op(v, dpp<0xB1>(v));    // quad_perm [1,0,3,2]
  v = op(v, dpp<0x4E>(v));    // quad_perm [2,3,0,1]
  v = op(v, dpp<0x124>(v));   // row_ror:4
  v = op(v, dpp<0x128>(v));   // row_ror:8
  return op(op(bc(v, 0), bc(v, 16)), op(bc(v, 32), bc(v, 48)));
}
template <typename T> __device__ inline T wave_sum(T v) { return wave_reduce(v, [](T a, T b) { return a + b; }); }
template <typename T> __device__ inline T wave_max(T v) { return wave_reduce(v, [](T a, T b) { return a > b ? a : b; }); }
template <typename T> __device__ inline T wave_min(T v) { return wave_reduce(v, [](T a, T b) { return a < b ? a : b; }); }
__device__ inline float  tdiv(float a, float b)  { return __fdividef(a, b); }   // a * rcp(b): ~1 ulp, QP arithmetic only
__device__ inline double tdiv(double a, double b) { return a / b; }
__device__ inline float  texp(float x)  { return __expf(x); }
__device__ inline double texp(double x) { return exp(x); }
__device__ inline float  trsqrt(float x)  { return rsqrtf(x); }
__device__ inline double trsqrt(double x) { return 1.0 / sqrt(x); }
__device__ inline float  tabs(float x)  { return fabsf(x); }
__device__ inline double tabs(double x) { return fabs(x); }

// AB' is stored with an XOR swizzle of its 4-float column groups, element (r, c) at
//   r*ABW + (((c >> 2) ^ sw(r)) << 2) + (c & 3),  sw(r) = (r >> 1) & 3,
// so that both access directions are (nearly) bank-conflict free: a fixed row read across 14 column
// lanes stays a permutation of one 16-float row, and a fixed column read across 13 row lanes spreads
// over 8 banks instead of 2.
__device__ inline int sw(int r) { return (r >> 1) & 3; }
__device__ inline int abo(int r, int c) { return r * ABW + ((((c >> 2) ^ sw(r)) << 2) | (c & 3)); }

template <typename T> __device__ inline void rotmat(const T* q, T* R) {
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz);     R[2] = 2 * (qx * qz + qw * qy);
  R[3] = 2 * (qx * qy + qw * qz);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
  R[6] = 2 * (qx * qz - qw * qy);     R[7] = 2 * (qy * qz + qw * qx);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

// quad constants in the arithmetic type of the caller
template <typename T> struct QC {
  T mass, J[3], tmax, xf[4], yf[4], zl[4], g;
  template <typename M> __device__ inline explicit QC(const M& m) {
    mass = (T)m.mass; tmax = (T)m.tmax; g = (T)m.g;
#pragma unroll
    for (int i = 0; i < 3; ++i) J[i] = (T)m.J[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xf[i] = (T)m.xf[i]; yf[i] = (T)m.yf[i]; zl[i] = (T)m.zl[i]; }
  }
};

// f(x,u) of the OCP model (src/quad_opt.py:186-251 in the reference); when `sub` != nullptr also
// writes the record the sensitivity pass needs: x(13) | d vdot/dq (3x4) | d vdot/dv (3x3) | R[:,2].
// GP term: m_d(s) = sum_j alpha_dj sf2 exp(-(s - X_j)^2 L2inv / 2), alpha = Kx^-1 mu.
template <typename T, typename TG>
__device__ inline void model_eval(const QC<T>& m, int nb, const TG* L2inv, const TG* sf2, const T* x, const T* u,
                                  const TG* alpha, const TG* basis, T* f, T* sub) {
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
      TG s0 = 0, s1 = 0;
      const TG vbd = (TG)vb[d];
      for (int j = 0; j < nb; ++j) {
        const TG dlt = vbd - basis[d * nb + j];
        const TG k = alpha[d * nb + j] * sf2[d] * texp(TG(-0.5) * dlt * dlt * L2inv[d]);
        s0 += k;
        s1 -= k * dlt;
      }
      mg[d] = (T)s0;
      mp[d] = (T)(s1 * L2inv[d]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) f[7 + i] += R[3 * i] * mg[0] + R[3 * i + 1] * mg[1] + R[3 * i + 2] * mg[2];
  }
  if (!sub) return;
#pragma unroll
  for (int i = 0; i < NX; ++i) sub[i] = x[i];
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
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

// one RK4 step of the NOMINAL model in double (quad_optimizer.discrete_dynamics on quad_nominal)
template <typename M>
__device__ inline void rk4_nominal(const M& m, const double* x, const double* u, double dt, double* xo) {
  const QC<double> qc(m);
  double k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
  const double* nul = nullptr;
  model_eval<double, double>(qc, 0, nul, nul, x, u, nul, nul, k1, (double*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k1[i];
  model_eval<double, double>(qc, 0, nul, nul, xt, u, nul, nul, k2, (double*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k2[i];
  model_eval<double, double>(qc, 0, nul, nul, xt, u, nul, nul, k3, (double*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
  model_eval<double, double>(qc, 0, nul, nul, xt, u, nul, nul, k4, (double*)nullptr);
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

// plant with drag (Quadrotor3D.f_nominal, drag=True, payload=False; src/quad.py:256-381), double
template <typename M>
__device__ inline void plant_eval(const M& m, const double* x, const double* u, double* f) {
  const double* q = x + 3; const double* v = x + 7; const double* r = x + 10;
  double R[9];
  rotmat(q, R);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  f[3] = 0.5 * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = 0.5 * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = 0.5 * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = 0.5 * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  const double aT = m.tmax * (u[0] + u[1] + u[2] + u[3]) / m.mass;
  double vb[3], ad[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sg = (double)((vb[i] > 0) - (vb[i] < 0));
    ad[i] = -m.aero_drag * vb[i] * vb[i] * sg / m.mass - m.rotor_drag[i] * vb[i] / m.mass;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) f[7 + i] = R[3 * i] * ad[0] + R[3 * i + 1] * ad[1] + R[3 * i + 2] * (ad[2] + aT);
  f[9] -= m.g;
  double ty = 0, tx = 0, tz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ty += u[j] * m.yf[j]; tx += u[j] * m.xf[j]; tz += u[j] * m.zl[j]; }
  f[10] = (m.tmax * ty + (m.J[1] - m.J[2]) * r[1] * r[2]) / m.J[0];
  f[11] = (-m.tmax * tx + (m.J[2] - m.J[0]) * r[2] * r[0]) / m.J[1];
  f[12] = (m.tmax * tz + (m.J[0] - m.J[1]) * r[0] * r[1]) / m.J[2];
}
template <typename M>
__device__ inline void plant_rk4(const M& m, double* x, const double* uin, double dt) {
  double u[4], k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = tmin(1.0, tmax(0.0, uin[j]));
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
// pass 1: lane per interval, 4 RK substages in TQ; writes records + gap c_i = Phi_i - X_{i+1}
// (the part X_i - X_{i+1} of the gap is formed in double)
template <typename TQ>
__device__ inline void shoot_states(const DevModel<TQ>& m, const double* D, TQ* S, const Lds& L, bool gp) {
  const int N = m.N;
  const QC<TQ> qc(m);
  const TQ h = (TQ)m.h;
  for (int i = threadIdx.x; i < N; i += 64) {
    TQ x[NX], u[NU], k[NX], xt[NX], acc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = (TQ)D[L.X + i * NX + j];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = (TQ)D[L.U + i * NU + j];
    const TQ* al = gp ? S + L.alpha : nullptr;
    TQ* sub = S + L.sub + i * SUBS;
    model_eval<TQ, TQ>(qc, m.nb, m.L2inv, m.sf2, x, u, al, S + L.basis, k, sub);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = k[j]; xt[j] = x[j] + h / 2 * k[j]; }
    model_eval<TQ, TQ>(qc, m.nb, m.L2inv, m.sf2, xt, u, al, S + L.basis, k, sub + SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] += 2 * k[j]; xt[j] = x[j] + h / 2 * k[j]; }
    model_eval<TQ, TQ>(qc, m.nb, m.L2inv, m.sf2, xt, u, al, S + L.basis, k, sub + 2 * SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] += 2 * k[j]; xt[j] = x[j] + h * k[j]; }
    model_eval<TQ, TQ>(qc, m.nb, m.L2inv, m.sf2, xt, u, al, S + L.basis, k, sub + 3 * SUBW);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const double gap = (D[L.X + i * NX + j] - D[L.X + (i + 1) * NX + j]) + (double)(h / 6 * (acc[j] + k[j]));
      S[L.c + i * NX + j] = (TQ)gap;
    }
  }
}
// pass 2: item = (interval i, column j of [A|B], j = 3..16) -> AB'[i][r][j-3]
template <typename TQ>
__device__ inline void shoot_sens(const DevModel<TQ>& m, TQ* S, const Lds& L) {
  const int N = m.N;
  const QC<TQ> qc(m);
  const TQ h = (TQ)m.h;
  const TQ a_s[4] = {TQ(0), TQ(0.5), TQ(0.5), TQ(1)};
  const TQ w_s[4] = {TQ(1), TQ(2), TQ(2), TQ(1)};
  const TQ c10 = (qc.J[1] - qc.J[2]) / qc.J[0], c11 = (qc.J[2] - qc.J[0]) / qc.J[1], c12 = (qc.J[0] - qc.J[1]) / qc.J[2];
  const TQ tm = qc.tmax / qc.mass;
  for (int it = threadIdx.x; it < N * 14; it += 64) {
    const int i = it / 14, jp = it - i * 14, j = 3 + jp;
    const bool ucol = j >= NX;
    TQ Sp[NX], acc[NX], Z[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) { Sp[r] = 0; acc[r] = 0; }
    TQ jur[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < NU; ++c)
      if (j - NX == c) { jur[0] = qc.tmax * qc.yf[c] / qc.J[0]; jur[1] = -qc.tmax * qc.xf[c] / qc.J[1]; jur[2] = qc.tmax * qc.zl[c] / qc.J[2]; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const TQ* sub = S + L.sub + i * SUBS + s * SUBW;
      const TQ hs = h * a_s[s];
#pragma unroll
      for (int r = 0; r < NX; ++r) Z[r] = ((r == j) ? TQ(1) : TQ(0)) + hs * Sp[r];
      const TQ qw = sub[3], qx = sub[4], qy = sub[5], qz = sub[6];
      const TQ r0 = sub[10], r1 = sub[11], r2 = sub[12];
      TQ Sn[NX];
      Sn[0] = Z[7]; Sn[1] = Z[8]; Sn[2] = Z[9];
      Sn[3] = TQ(0.5) * (-r0 * Z[4] - r1 * Z[5] - r2 * Z[6] - qx * Z[10] - qy * Z[11] - qz * Z[12]);
      Sn[4] = TQ(0.5) * (r0 * Z[3] + r2 * Z[5] - r1 * Z[6] + qw * Z[10] - qz * Z[11] + qy * Z[12]);
      Sn[5] = TQ(0.5) * (r1 * Z[3] - r2 * Z[4] + r0 * Z[6] + qz * Z[10] + qw * Z[11] - qx * Z[12]);
      Sn[6] = TQ(0.5) * (r2 * Z[3] + r1 * Z[4] - r0 * Z[5] - qy * Z[10] + qx * Z[11] + qw * Z[12]);
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        TQ t = ucol ? sub[34 + row] * tm : TQ(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) t += sub[13 + row * 4 + c] * Z[3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) t += sub[25 + row * 3 + c] * Z[7 + c];
        Sn[7 + row] = t;
      }
      Sn[10] = jur[0] + c10 * (r2 * Z[11] + r1 * Z[12]);
      Sn[11] = jur[1] + c11 * (r2 * Z[10] + r0 * Z[12]);
      Sn[12] = jur[2] + c12 * (r1 * Z[10] + r0 * Z[11]);
#pragma unroll
      for (int r = 0; r < NX; ++r) { acc[r] += w_s[s] * Sn[r]; Sp[r] = Sn[r]; }
    }
    TQ* AB = S + L.AB + i * ABS;
#pragma unroll
    for (int r = 0; r < NX; ++r) AB[abo(r, jp)] = ((r == j) ? TQ(1) : TQ(0)) + h / 6 * acc[r];
  }
  // zero the two pad columns (read by the vectorised 4-wide loads)
  for (int it = threadIdx.x; it < N * NX * 2; it += 64) {
    const int row = it >> 1, i = row / NX, r = row - i * NX;
    S[L.AB + i * ABS + abo(r, 14 + (it & 1))] = 0;
  }
}

// ------------------------------------------------------------------ QP: vector sweeps
// QP in (dx, du): min sum_i 1/2 dx'Q_i dx + qv_i'dx + 1/2 du'R du + r0_i'du  s.t. dx_{i+1} = A dx_i + B du_i + c_i,
// dx_0 = x0 - X_0, lb <= du <= ub.  Q_i = h W_x (i<N) / W_e, R = h W_u diagonal; qv, r0, lb, ub, c prepared by
// the caller.  Vectors of length 13 live on lanes 0..12, input-sized quantities on lanes 13..16.

// column `col` of a stage's AB' (13 values, one per row k), into registers
template <typename TQ> __device__ inline void load_col(const TQ* ab, int col, TQ (&o)[NX]) {
#pragma unroll
  for (int k = 0; k < NX; ++k) o[k] = ab[abo(k, col)];
}
// row r of a stage's AB' (14 values + 2 pads) into registers, logical column order
template <typename TQ> __device__ inline void load_row(const TQ* ab, int r, TQ (&o)[16]) {
  const int s = sw(r);
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = ab[r * ABW + ((((c >> 2) ^ s) << 2) | (c & 3))];
}

// forward rollout dx_{i+1} = A dx_i + B z_i (+ c_i); dx_0 taken from S[dxo + 0..12]
template <typename TQ>
__device__ inline void rollout(const DevModel<TQ>& m, TQ* S, const Lds& L, int dxo, int zo, bool with_c) {
  const int N = m.N, lane = threadIdx.x, r = lane < NX ? lane : 0;
  TQ xr = S[dxo + r];
  TQ cur[16], nxt[16];
  load_row(S + L.AB, r, cur);
  for (int i = 0; i < N; ++i) {
    if (i + 1 < N) load_row(S + L.AB + (i + 1) * ABS, r, nxt);
    TQ t = (with_c ? S[L.c + i * NX + r] : TQ(0)) + (r < 3 ? xr : TQ(0));
#pragma unroll
    for (int j = 0; j < NU; ++j) t += cur[10 + j] * S[zo + i * NU + j];
#pragma unroll
    for (int k = 3; k < NX; ++k) t += cur[k - 3] * bc(xr, k);
    xr = t;
    if (lane < NX) S[dxo + (i + 1) * NX + lane] = t;
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
  }
  __syncthreads();
}

// adjoint sweep: grad = d/dz of the QP objective at (dx(z), z)
template <typename TQ>
__device__ inline void adjoint(const DevModel<TQ>& m, TQ* S, const Lds& L) {
  const int N = m.N, lane = threadIdx.x;
  const int a = lane < NY ? lane : 0, col = a >= 3 ? a - 3 : 0, ax = a < NX ? a : 0, au = a >= NX ? a - NX : 0;
  const TQ qd = a < NX ? (TQ)(m.h * m.W[a]) : TQ(0), qe = a < NX ? (TQ)m.We[a] : TQ(0);
  const TQ rd = (a >= NX) ? (TQ)(m.h * m.W[a]) : TQ(0);
  TQ pi = a < NX ? qe * S[L.dx + N * NX + a] + S[L.qv + N * NX + a] : TQ(0);
  TQ cur[NX], nxt[NX];
  load_col(S + L.AB + (N - 1) * ABS, col, cur);
  for (int i = N - 1; i >= 0; --i) {
    if (i > 0) load_col(S + L.AB + (i - 1) * ABS, col, nxt);
    const TQ add = a < NX ? qd * S[L.dx + i * NX + ax] + S[L.qv + i * NX + ax]
                          : rd * S[L.z + i * NU + au] + S[L.r0 + i * NU + au];
    TQ t = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) t += cur[k] * bc(pi, k);
    if (a < 3) t = pi;
    t += add;
    if (lane < NX) pi = t;
    else if (lane < NY) S[L.grad + i * NU + au] = t;
#pragma unroll
    for (int k = 0; k < NX; ++k) cur[k] = nxt[k];
  }
  __syncthreads();
}

// backward vector recursion with stored K, Linv: feed-forward kv for linear term rho
template <typename TQ>
__device__ inline void riccati_backward_vec(const DevModel<TQ>& m, TQ* S, const Lds& L, bool polish) {
  const int N = m.N, lane = threadIdx.x;
  const int a = lane < NY ? lane : 0, col = a >= 3 ? a - 3 : 0, j = a >= NX ? a - NX : 0, b = a < NX ? a : 0;
  TQ pv = 0;
  TQ cur[NX], nxt[NX];
  load_col(S + L.AB + (N - 1) * ABS, col, cur);
  for (int i = N - 1; i >= 0; --i) {
    if (i > 0) load_col(S + L.AB + (i - 1) * ABS, col, nxt);
    // per-lane coefficients of the second half: K[:,b] for lanes < 13, Linv[j][:] for lanes 13..16
    const TQ* cf = lane < NX ? S + L.K + i * KS + b : S + L.Linv + i * 16 + j * 4;
    const int cs = lane < NX ? ABW : 1;
    const TQ c0 = cf[0], c1 = cf[cs], c2 = cf[2 * cs], c3 = cf[3 * cs];
    const TQ rho = S[L.rho + i * NU + j];
    TQ t = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) t += cur[k] * bc(pv, k);
    if (a < 3) t = pv;
    TQ gt = (a >= NX) ? rho + t : TQ(0);
    if (polish && a >= NX && S[L.act + i * NU + j] != TQ(0)) gt = 0;
    const TQ g0 = bc(gt, 13), g1 = bc(gt, 14), g2 = bc(gt, 15), g3 = bc(gt, 16);
    const TQ comb = c0 * g0 + c1 * g1 + c2 * g2 + c3 * g3;
    pv = lane < NX ? t + comb : TQ(0);
    if (a >= NX && lane < NY) S[L.kv + i * NU + j] = -comb;
#pragma unroll
    for (int k = 0; k < NX; ++k) cur[k] = nxt[k];
  }
  __syncthreads();
}

// forward sweep: Dx_0 = 0; dz_i = K_i Dx_i + k_i ; Dx_{i+1} = A Dx_i + B dz_i   (out: S[dzo], S[L.Dx])
template <typename TQ>
__device__ inline void riccati_forward(const DevModel<TQ>& m, TQ* S, const Lds& L, int dzo) {
  const int N = m.N, lane = threadIdx.x;
  const int r = lane < NX ? lane : 0, j = (lane >= NX && lane < NY) ? lane - NX : 0;
  TQ xr = 0;
  if (lane < NX) S[L.Dx + lane] = 0;
  // per-lane coefficient row over x_0..x_12: lanes < 13 -> row r of A (identity columns 0..2, then AB' cols 0..9),
  // lanes 13..16 -> row j of K
  TQ cur[NX], nxt[NX], bcur[NU], bnxt[NU];
  auto load = [&](int i, TQ (&cx)[NX], TQ (&cb)[NU]) {
    if (lane < NX) {
      TQ row[16];
      load_row(S + L.AB + i * ABS, r, row);
#pragma unroll
      for (int k = 0; k < 3; ++k) cx[k] = lane == k ? TQ(1) : TQ(0);
#pragma unroll
      for (int k = 3; k < NX; ++k) cx[k] = row[k - 3];
#pragma unroll
      for (int k = 0; k < NU; ++k) cb[k] = row[10 + k];
    } else {
      const TQ* kr = S + L.K + i * KS + j * ABW;
#pragma unroll
      for (int k = 0; k < NX; ++k) cx[k] = kr[k];
#pragma unroll
      for (int k = 0; k < NU; ++k) cb[k] = 0;
    }
  };
  load(0, cur, bcur);
  for (int i = 0; i < N; ++i) {
    if (i + 1 < N) load(i + 1, nxt, bnxt);
    TQ t = lane < NX ? TQ(0) : S[L.kv + i * NU + j];
#pragma unroll
    for (int k = 0; k < NX; ++k) t += cur[k] * bc(xr, k);
    const TQ d0 = bc(t, 13), d1 = bc(t, 14), d2 = bc(t, 15), d3 = bc(t, 16);
    if (lane >= NX && lane < NY) S[dzo + i * NU + j] = t;
    xr = lane < NX ? t + bcur[0] * d0 + bcur[1] * d1 + bcur[2] * d2 + bcur[3] * d3 : TQ(0);
    if (lane < NX) S[L.Dx + (i + 1) * NX + lane] = xr;
#pragma unroll
    for (int k = 0; k < NX; ++k) cur[k] = nxt[k];
#pragma unroll
    for (int k = 0; k < NU; ++k) bcur[k] = bnxt[k];
  }
  __syncthreads();
}

// ------------------------------------------------------------------ QP: Riccati factorisation
// Backward sweep recomputing P_i, K_i, Lambda_i^-1 for R~ = R + (polish ? 0 : ll/sl + lu/su) with inputs
// pinned by `act` eliminated in polish mode, merged with the vector recursion for the linear term rho.
// Uses the structure A[:,0:3] = [I;0]:  with T1' = P [A|B]' (cols 3..16), the blocks of [A B]^T P [A B] are
//   [a<3][b<3] = P[a][b],  [a<3][b>=3] = T1'[a][b-3],  [a>=3][b>=3] = F'[a-3][b-3] = sum_k AB'[k][a-3] T1'[k][b-3].
// Returns false if a stage Hessian was not positive definite.
template <typename TQ>
__device__ inline bool riccati_factor(const DevModel<TQ>& m, TQ* S, const Lds& L, bool polish) {
  const int N = m.N, lane = threadIdx.x, nv = N * NU;
  // ---- per-lane roles, fixed over the sweep
  const int r4 = lane >> 2, g4 = lane & 3;                       // Ph1/Ph4: (row, 4-column group), lanes 0..51
  int fa = 0, fg = 0;                                            // Ph2: lane -> (a', g) with 4g <= a' (32 items)
  {
    int cnt = 0;
    for (int a = 0; a < 14; ++a)
      for (int g = 0; g * 4 <= a; ++g) { if (cnt == lane) { fa = a; fg = g; } ++cnt; }
  }
  const int va = lane < NY ? lane : 0, vcol = va >= 3 ? va - 3 : 0, vj = va >= NX ? va - NX : 0;
  int goff[4];   // Ph4 gather offsets for G(r, c) = ([A B]^T P [A B])[r][c], c = 4*g4 + cc
  bool gisP[4];  // source is P_next (ping-pong base added at use)
  int moff[4];   // M[j][r4] source
  int m3off[4];  // Ph3 lane b < 13: M[:,b] sources
  bool ok = true;
  {
    const int r = r4 < NX ? r4 : 0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      int c = 4 * g4 + cc;
      if (c >= NX) c = NX - 1;
      gisP[cc] = (r < 3 && c < 3);
      if (r < 3 && c < 3) goff[cc] = r * ABW + c;
      else if (r < 3) goff[cc] = L.T1 + r * ABW + (c - 3);
      else if (c < 3) goff[cc] = L.T1 + c * ABW + (r - 3);
      else goff[cc] = L.F + ((r > c ? r : c) - 3) * ABW + ((r > c ? c : r) - 3);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) moff[j] = r < 3 ? L.T1 + r * ABW + 10 + j : L.F + (10 + j) * ABW + (r - 3);
    const int b = lane < NX ? lane : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) m3off[j] = b < 3 ? L.T1 + b * ABW + 10 + j : L.F + (10 + j) * ABW + (b - 3);
  }
  const TQ qdiag = r4 < NX ? (TQ)(m.h * m.W[r4]) : TQ(0);
  // stage input Hessian diagonals R~ (negative value = input pinned by the polish)
  for (int i = lane; i < nv; i += 64) {
    const TQ rr = (TQ)(m.h * m.W[NX + (i & 3)]);
    TQ v;
    if (!polish) v = rr + tdiv(S[L.ll + i], S[L.sl + i]) + tdiv(S[L.lu + i], S[L.su + i]);
    else v = S[L.act + i] != TQ(0) ? TQ(-1) : rr;
    S[L.rt + i] = v;
  }
  // P_N = W_e ; K pads
  for (int it = lane; it < ABS; it += 64) S[L.P + it] = ((it >> 4) == (it & 15) && (it >> 4) < NX) ? (TQ)m.We[it >> 4] : TQ(0);
  for (int it = lane; it < N * NU * 3; it += 64) S[L.K + (it / 3) * ABW + NX + it % 3] = 0;
  TQ pv = 0;
  TQ vc[NX], vn[NX];
  load_col(S + L.AB + (N - 1) * ABS, vcol, vc);
  __syncthreads();
  for (int i = N - 1; i >= 0; --i) {
    const int pc = (N - 1 - i) & 1;
    const TQ* AB = S + L.AB + i * ABS;
    const TQ* Pn = S + L.P + pc * ABS;
    if (i > 0) load_col(S + L.AB + (i - 1) * ABS, vcol, vn);
    // ---- Ph1: T1' = P_{i+1} AB'   (13 x 14, 4 columns per lane) ; vector t = [A B]^T p_{i+1}
    if (lane < 52) {
      TQ prow[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<TQ> v = *reinterpret_cast<const V4<TQ>*>(Pn + r4 * ABW + 4 * q);
        prow[4 * q] = v.a; prow[4 * q + 1] = v.b; prow[4 * q + 2] = v.c; prow[4 * q + 3] = v.d;
      }
      TQ a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const V4<TQ> v = *reinterpret_cast<const V4<TQ>*>(AB + k * ABW + ((g4 ^ sw(k)) << 2));
        a0 += prow[k] * v.a; a1 += prow[k] * v.b; a2 += prow[k] * v.c; a3 += prow[k] * v.d;
      }
      V4<TQ> o; o.a = a0; o.b = a1; o.c = a2; o.d = a3;
      *reinterpret_cast<V4<TQ>*>(S + L.T1 + r4 * ABW + 4 * g4) = o;
    }
    TQ t = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) t += vc[k] * bc(pv, k);
    if (va < 3) t = pv;
    __syncthreads();
    // ---- Ph2: F' lower block-triangle = AB'^T T1'
    if (lane < 32) {
      TQ a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const TQ p = AB[abo(k, fa)];
        const V4<TQ> v = *reinterpret_cast<const V4<TQ>*>(S + L.T1 + k * ABW + 4 * fg);
        a0 += p * v.a; a1 += p * v.b; a2 += p * v.c; a3 += p * v.d;
      }
      V4<TQ> o; o.a = a0; o.b = a1; o.c = a2; o.d = a3;
      *reinterpret_cast<V4<TQ>*>(S + L.F + fa * ABW + 4 * fg) = o;
    }
    __syncthreads();
    // ---- Ph3: Lambda = R~ + F_uu, Cholesky in registers (redundantly on every lane),
    //           lanes b<13: K[:,b] = -Lambda^-1 M[:,b]; lanes 13..16: column of Lambda^-1
    {
      TQ Lm[4][4];
      bool am[4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 <= a; ++b2) Lm[a][b2] = S[L.F + (10 + a) * ABW + 10 + b2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const TQ rt = S[L.rt + i * NU + a];
        am[a] = rt < TQ(0);
        Lm[a][a] += rt;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (am[a]) {
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) { if (b2 < a) Lm[a][b2] = 0; if (b2 > a) Lm[b2][a] = 0; }
          Lm[a][a] = 1;
        }
      TQ id[4];
      bool pd = true;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        TQ d = Lm[c][c];
#pragma unroll
        for (int k = 0; k < c; ++k) d -= Lm[c][k] * Lm[c][k];
        if (!(d > 0)) { pd = false; d = 1; }
        id[c] = trsqrt(d);
#pragma unroll
        for (int a = c + 1; a < 4; ++a) {
          TQ s2 = Lm[a][c];
#pragma unroll
          for (int k = 0; k < c; ++k) s2 -= Lm[a][k] * Lm[c][k];
          Lm[a][c] = s2 * id[c];
        }
      }
      if (!pd) ok = false;
      // rhs: M[:,b] for lanes < 13, e_j for lanes 13..16
      TQ y[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = lane < NX ? (am[c] ? TQ(0) : S[m3off[c]]) : ((lane - NX) == c ? TQ(1) : TQ(0));
      // forward solve G y' = y, backward solve G^T x = y'
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        TQ s2 = y[c];
#pragma unroll
        for (int k = 0; k < c; ++k) s2 -= Lm[c][k] * y[k];
        y[c] = s2 * id[c];
      }
#pragma unroll
      for (int c = 3; c >= 0; --c) {
        TQ s2 = y[c];
#pragma unroll
        for (int k = c + 1; k < 4; ++k) s2 -= Lm[k][c] * y[k];
        y[c] = s2 * id[c];
      }
      // vector part: gt_j on lanes 13..16, broadcast
      TQ gt = (va >= NX) ? S[L.rho + i * NU + vj] + t : TQ(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (va >= NX && vj == c && am[c]) gt = 0;
      const TQ g0 = bc(gt, 13), g1 = bc(gt, 14), g2 = bc(gt, 15), g3 = bc(gt, 16);
      if (lane < NX) {
        TQ* Kc = S + L.K + i * KS + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) { if (am[c]) y[c] = 0; Kc[c * ABW] = -y[c]; }
        pv = t - (y[0] * g0 + y[1] * g1 + y[2] * g2 + y[3] * g3);
      } else {
        if (lane < NY) {
          TQ* Li = S + L.Linv + i * 16 + vj * 4;
#pragma unroll
          for (int c = 0; c < 4; ++c) Li[c] = y[c];
          TQ kvv = -(y[0] * g0 + y[1] * g1 + y[2] * g2 + y[3] * g3);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (vj == c && am[c]) kvv = 0;
          S[L.kv + i * NU + vj] = kvv;
        }
        pv = 0;
      }
    }
    if (i == 0) break;
    __syncthreads();
    // ---- Ph4: P_i = Q + A^T P A + M^T K   (13 x 13, 4 columns per lane)
    if (lane < 52) {
      TQ mj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) mj[j] = S[moff[j]];
      TQ o[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) o[cc] = S[goff[cc] + (gisP[cc] ? L.P + pc * ABS : 0)];
      const TQ* Kr = S + L.K + i * KS + 4 * g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const V4<TQ> kv4 = *reinterpret_cast<const V4<TQ>*>(Kr + j * ABW);
        o[0] += mj[j] * kv4.a; o[1] += mj[j] * kv4.b; o[2] += mj[j] * kv4.c; o[3] += mj[j] * kv4.d;
      }
      const int cd = r4 - 4 * g4;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (cd == cc) o[cc] += qdiag;
      V4<TQ> ov; ov.a = o[0]; ov.b = o[1]; ov.c = o[2]; ov.d = o[3];
      *reinterpret_cast<V4<TQ>*>(S + L.P + (1 - pc) * ABS + r4 * ABW + 4 * g4) = ov;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) vc[k] = vn[k];
    __syncthreads();
  }
  __syncthreads();
  return wave_min<int>(ok ? 1 : 0) != 0;   // all lanes agree on definiteness
}

// ------------------------------------------------------------------ QP: IPM + polish
// Mehrotra predictor-corrector iterations; every Newton system is one Riccati factorisation + two
// vector sweeps.  Continues from the current (z, sl, su, ll, lu, dx, grad) until |r_d| <= tol*gm and
// mu <= tol.  returns 0 converged / 1 NaN / 2 iteration cap / 4 stage Hessian not positive definite
template <typename TQ>
__device__ inline int ipm_run(const DevModel<TQ>& m, TQ* S, const Lds& L, const TQ tol, const TQ gm, int& it PF_ARG) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x;
  int status = 2;
  const int maxit = m.qp_max_iter;
  for (; it < maxit; ++it) {
    TQ rdm = 0, mu = 0;
    for (int i = tid; i < nv; i += 64) {
      rdm = tmax(rdm, tabs(S[L.grad + i] - S[L.ll + i] + S[L.lu + i]));
      mu += S[L.sl + i] * S[L.ll + i] + S[L.su + i] * S[L.lu + i];
    }
    rdm = wave_max(rdm);
    mu = wave_sum(mu) / (2 * nv);
    if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
    if (rdm <= tol * gm && mu <= tol) { status = 0; break; }
    // predictor: (H + Sigma) dza = -grad
    for (int i = tid; i < nv; i += 64) S[L.rho + i] = S[L.grad + i];
    __syncthreads();
    PF_START();
    const bool fok = riccati_factor(m, S, L, false);
    PF_STOP(PF_FACTOR);
    if (!fok) { status = 4; break; }
    PF_START(); riccati_forward(m, S, L, L.dza); PF_STOP(PF_FWD);
    TQ aff = 1;
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dl = -ll - tdiv(ll, sl) * d, du = -lu + tdiv(lu, su) * d;
      if (d < 0) aff = tmin(aff, tdiv(-sl, d));
      if (d > 0) aff = tmin(aff, tdiv(su, d));
      if (dl < 0) aff = tmin(aff, tdiv(-ll, dl));
      if (du < 0) aff = tmin(aff, tdiv(-lu, du));
    }
    aff = wave_min(aff);
    TQ mua = 0;
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dl = -ll - tdiv(ll, sl) * d, du = -lu + tdiv(lu, su) * d;
      mua += (sl + aff * d) * (ll + aff * dl) + (su - aff * d) * (lu + aff * du);
    }
    mua = wave_sum(mua) / (2 * nv);
    TQ sigma = mua / mu;
    sigma = sigma * sigma * sigma;
    // corrector rhs r = -rd + rcl/sl - rcu/su ; linear term rho = -r
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dl = -ll - tdiv(ll, sl) * d, du = -lu + tdiv(lu, su) * d;
      const TQ rcl = -sl * ll + sigma * mu - d * dl;
      const TQ rcu = -su * lu + sigma * mu + d * du;
      const TQ rd = S[L.grad + i] - ll + lu;
      S[L.rho + i] = rd - tdiv(rcl, sl) + tdiv(rcu, su);
    }
    __syncthreads();
    PF_START(); riccati_backward_vec(m, S, L, false); PF_STOP(PF_BWD);
    PF_START(); riccati_forward(m, S, L, L.dz); PF_STOP(PF_FWD);
    TQ ap = 1, ad = 1;
    for (int i = tid; i < nv; i += 64) {
      const TQ da = S[L.dza + i], d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dla = -ll - tdiv(ll, sl) * da, dua = -lu + tdiv(lu, su) * da;
      const TQ rcl = -sl * ll + sigma * mu - da * dla;
      const TQ rcu = -su * lu + sigma * mu + da * dua;
      const TQ dl = tdiv(rcl - ll * d, sl), du = tdiv(rcu + lu * d, su);
      if (d < 0) ap = tmin(ap, tdiv(-sl, d));
      if (d > 0) ap = tmin(ap, tdiv(su, d));
      if (dl < 0) ad = tmin(ad, tdiv(-ll, dl));
      if (du < 0) ad = tmin(ad, tdiv(-lu, du));
    }
    ap = wave_min(ap);
    ad = wave_min(ad);
    const TQ tau = tmax(TQ(0.995), 1 - mu);
    ap = tmin(TQ(1), tau * ap);
    ad = tmin(TQ(1), tau * ad);
    for (int i = tid; i < nv; i += 64) {
      const TQ da = S[L.dza + i], d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dla = -ll - tdiv(ll, sl) * da, dua = -lu + tdiv(lu, su) * da;
      const TQ rcl = -sl * ll + sigma * mu - da * dla;
      const TQ rcu = -su * lu + sigma * mu + da * dua;
      const TQ dl = tdiv(rcl - ll * d, sl), du = tdiv(rcu + lu * d, su);
      S[L.z + i] += ap * d; S[L.sl + i] = sl + ap * d; S[L.su + i] = su - ap * d;
      S[L.ll + i] = ll + ad * dl; S[L.lu + i] = lu + ad * du;
    }
    for (int i = tid; i < (N + 1) * NX; i += 64) S[L.dx + i] += ap * S[L.Dx + i];
    __syncthreads();
    PF_START(); adjoint(m, S, L); PF_STOP(PF_ADJ);
  }
  return status;
}

// Active-set polish: starting from the IPM point, pin the inputs the IPM identifies as active and
// take Newton steps on the free set (masked Riccati) with a ratio test; inputs whose multiplier has
// the wrong sign are released (the worst one, only at a minimiser of the current working set), inputs
// that block are pinned.  Ends on an exact KKT point of the QP (to rounding), which an interior
// method only approaches like sqrt(mu) on weakly active bounds.
template <typename TQ>
__device__ inline bool polish(const DevModel<TQ>& m, TQ* S, const Lds& L, const TQ gm, int& passes PF_ARG) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x;
  for (int i = tid; i < nv; i += 64)
    S[L.act + i] = S[L.ll + i] > S[L.sl + i] ? TQ(-1) : (S[L.lu + i] > S[L.su + i] ? TQ(1) : TQ(0));
  __syncthreads();
  const TQ tolm = 64 * m.eps * gm;  // multiplier sign / stationarity
  const TQ tolb = 16 * m.eps;       // bound proximity (bounds are O(1))
  bool refactor = true, settled = false, full = false;
  TQ gF_prev = TQ(1e30);
  for (passes = 0; passes < m.polish_max; ++passes) {
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i];
      if (a < 0) S[L.z + i] = S[L.lb + i];
      else if (a > 0) S[L.z + i] = S[L.ub + i];
    }
    __syncthreads();
    PF_START(); rollout(m, S, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
    PF_START(); adjoint(m, S, L); PF_STOP(PF_ADJ);
    // stationarity on the free set, worst multiplier sign violation on the pinned set
    TQ gF = 0, vmax = 0;
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i], g = S[L.grad + i];
      if (a == TQ(0)) gF = tmax(gF, tabs(g));
      else vmax = tmax(vmax, a < 0 ? -g : g);
    }
    gF = wave_max(gF);
    vmax = wave_max(vmax);
    if (!(gF == gF)) return false;
    if (full) {
      // the point minimises the QP on the working set: multipliers are meaningful here only
      if (vmax > tolm) {
        for (int i = tid; i < nv; i += 64) {
          const TQ a = S[L.act + i], g = S[L.grad + i];
          if (a != TQ(0) && (a < 0 ? -g : g) >= vmax) S[L.act + i] = 0;  // release the worst one
        }
        refactor = true;
        gF_prev = TQ(1e30);
        __syncthreads();
      } else if (gF <= tolm || gF > TQ(0.25) * gF_prev) {
        settled = true;  // stationary, or refinement stagnated at the rounding level
        break;
      } else {
        gF_prev = gF;
      }
    }
    for (int i = tid; i < nv; i += 64) S[L.rho + i] = S[L.grad + i];
    __syncthreads();
    PF_START();
    if (refactor) { const bool fok = riccati_factor(m, S, L, true); PF_STOP(PF_FACTOR); if (!fok) return false; }
    else { riccati_backward_vec(m, S, L, true); PF_STOP(PF_BWD); }
    refactor = false;
    PF_START(); riccati_forward(m, S, L, L.dz); PF_STOP(PF_FWD);
    TQ alpha = 1;
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const TQ d = S[L.dz + i], z = S[L.z + i];
      if (d < 0) alpha = tmin(alpha, tmax(TQ(0), (S[L.lb + i] - z) / d));
      if (d > 0) alpha = tmin(alpha, tmax(TQ(0), (S[L.ub + i] - z) / d));
    }
    alpha = wave_min(alpha);
    int nblk = 0;
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const TQ d = S[L.dz + i], z = S[L.z + i] + alpha * d;
      S[L.z + i] = z;
      if (alpha < TQ(1)) {
        if (d < 0 && z <= S[L.lb + i] + tolb) { S[L.act + i] = -1; nblk += 1; }
        else if (d > 0 && z >= S[L.ub + i] - tolb) { S[L.act + i] = 1; nblk += 1; }
      }
    }
    nblk = wave_sum(nblk);
    full = nblk == 0;
    if (nblk > 0) refactor = true;
    __syncthreads();
  }
  return settled;
}

// Box-QP solve: IPM to the hand-over tolerance, then active-set polish; if the polish does not
// settle (degenerate cycling), fall back to IPM iterations down to the final tolerance.
// On exit S[L.z] holds the solution and S[L.dx] the matching state trajectory; returns passes.
template <typename TQ>
__device__ inline int solve_qp(const DevModel<TQ>& m, TQ* S, const Lds& L, int* status PF_ARG) {
  const int N = m.N, nv = N * NU, tid = threadIdx.x;
  // interior start
  for (int i = tid; i < nv; i += 64) {
    const TQ lb = S[L.lb + i], ub = S[L.ub + i], w = ub - lb;
    const TQ z0 = tmin(tmax(TQ(0), lb + TQ(0.1) * w), ub - TQ(0.1) * w);
    S[L.z + i] = z0; S[L.sl + i] = z0 - lb; S[L.su + i] = ub - z0;
  }
  __syncthreads();
  PF_START(); rollout(m, S, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
  PF_START(); adjoint(m, S, L); PF_STOP(PF_ADJ);
  TQ gm = 1;
  for (int i = tid; i < nv; i += 64) gm = tmax(gm, tabs(S[L.grad + i]));
  gm = wave_max(gm);
  for (int i = tid; i < nv; i += 64) { S[L.ll + i] = TQ(0.1) * gm / S[L.sl + i]; S[L.lu + i] = TQ(0.1) * gm / S[L.su + i]; }
  __syncthreads();
  int it = 0, passes = 0;
  int st = ipm_run(m, S, L, m.polish_max > 0 ? m.ipm_tol : m.qp_tol, gm, it PF_PASS);
  if (st == 0 && m.polish_max > 0) {
    for (int i = tid; i < nv; i += 64) S[L.dza + i] = S[L.z + i];
    __syncthreads();
    if (!polish(m, S, L, gm, passes PF_PASS)) {
      for (int i = tid; i < nv; i += 64) S[L.z + i] = S[L.dza + i];
      __syncthreads();
      PF_START(); rollout(m, S, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
      PF_START(); adjoint(m, S, L); PF_STOP(PF_ADJ);
      st = ipm_run(m, S, L, m.qp_tol, gm, it PF_PASS);
    }
  }
  PF_START(); rollout(m, S, L, L.dx, L.z, true); PF_STOP(PF_ROLL);   // state trajectory of the returned z
  *status = st;
  return it + passes;
}

// ------------------------------------------------------------------ RGP regress (3 axes, one new point each)
// RGP.regress / RGP.predict of the reference (src/gp/RGP.py:199-208,303-330), scalar new point:
//   J = k* Kx^-1 ; mu_p = J mu ; Cp = sf2 - J k* + J C J^T ; G = C J^T/(Cp + sn2) ;
//   mu += G (y - mu_p) ; C -= G (J C)      (not symmetrised, as in the reference)
template <typename TQ>
__device__ inline void rgp_regress(const DevModel<TQ>& m, TQ* S, const Lds& L, TQ* gmu, TQ* gC, const double* vb, const double* ad) {
  const int n = m.nb, tid = threadIdx.x, NT = blockDim.x, n3 = 3 * n, nn = n * n;
  TQ* C = S + L.rgp;
  TQ* ks = C + al4(3 * nn);
  TQ* Jt = ks + al4(n3);
  TQ* JC = Jt + al4(n3);
  TQ* CJ = JC + al4(n3);
  TQ* mu = CJ + al4(n3);
  TQ* sc = mu + al4(n3);
  for (int i = tid; i < 3 * nn; i += NT) C[i] = gC[i];
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n;
    const TQ dl = (TQ)vb[d] - m.basis[i];
    ks[i] = m.sf2[d] * texp(TQ(-0.5) * dl * dl * m.L2inv[d]);
    mu[i] = gmu[i];
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    TQ t = 0;
    for (int k = 0; k < n; ++k) t += ks[d * n + k] * m.Kxinv[d * nn + k * n + j];
    Jt[i] = t;
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    TQ t = 0, s = 0;
    for (int k = 0; k < n; ++k) { t += Jt[d * n + k] * C[d * nn + k * n + j]; s += C[d * nn + j * n + k] * Jt[d * n + k]; }
    JC[i] = t;
    CJ[i] = s;
  }
  __syncthreads();
  if (tid < 3) {
    const int d = tid;
    TQ mup = 0, Jk = 0, JCJ = 0;
    for (int k = 0; k < n; ++k) { mup += Jt[d * n + k] * mu[d * n + k]; Jk += Jt[d * n + k] * ks[d * n + k]; JCJ += JC[d * n + k] * Jt[d * n + k]; }
    const TQ Cp = m.sf2[d] - Jk + JCJ;
    sc[d * 4] = ((TQ)ad[d] - mup);
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
// reference row (get_reference_chunk, src/utils/utils.py:897-931) for horizon node j
__device__ inline long chunk_row(int j, int have, int idx, int skip, int len) { return j < have ? (long)idx + (long)j * skip : (long)len - 1; }

template <typename TQ>
__global__ void __launch_bounds__(64) step_kernel(const DevModel<TQ> m, const DevState<TQ> st, const int mode) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = m.N, nb = m.nb, nv = N * NU;
  const Lds L = lds_layout(N, nb);
  double* D = reinterpret_cast<double*>(smem_raw);
  TQ* S = reinterpret_cast<TQ*>(smem_raw + L.dbytes);
  const bool gp = nb > 0;
#ifdef MPCQ_PROFILE
  Prof pf;
  for (int k = 0; k < PF_N; ++k) pf.acc[k] = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  pf.t = t_begin;
#endif
  // ---- load persistent state (lane-contiguous records)
  double* gX = st.X + (size_t)b * (N + 1) * NX;
  double* gU = st.U + (size_t)b * N * NU;
  for (int i = tid; i < (N + 1) * NX; i += 64) D[L.X + i] = gX[i];
  for (int i = tid; i < nv; i += 64) D[L.U + i] = gU[i];
  if (tid < NX) D[L.x0 + tid] = st.x_meas[(size_t)b * NX + tid];
  const int idx = st.idx[b];
  int have = 0, len = 1;
  const double* tr = nullptr;
  if (mode & MODE_TRAJ) {
    len = st.tlen[b];
    const long left = (long)len - idx;
    if (left > (long)N * m.skip) have = N;
    else if (left > m.skip - 1) { have = (int)((left + m.skip - 1) / m.skip); if (have > N) have = N; }
    tr = st.traj + (size_t)b * m.Tmax * NX;
  }
  const double* gy = st.yref + (size_t)b * N * NY;
  const double* gyN = st.yrefN + (size_t)b * NX;
  auto xref = [&](int i, int k) -> double {  // reference of node i (i == N: terminal = chunk row N-1)
    if (mode & MODE_TRAJ) return tr[chunk_row(i < N ? i : N - 1, have, idx, m.skip, len) * NX + k];
    return i < N ? gy[i * NY + k] : gyN[k];
  };
  auto uref = [&](int i, int k) -> double { return (mode & MODE_TRAJ) ? m.uref[k] : gy[i * NY + NX + k]; };
  __syncthreads();
  // QP data formed in double: qv = Q_i (X_i - xref_i), r0 = R (U_i - uref_i), bounds
  for (int it = tid; it < (N + 1) * NX; it += 64) {
    const int i = it / NX, k = it - i * NX;
    const double q = i < N ? m.h * m.W[k] : m.We[k];
    S[L.qv + it] = (TQ)(q * (D[L.X + it] - xref(i, k)));
  }
  for (int it = tid; it < nv; it += 64) {
    const int i = it >> 2, k = it & 3;
    const double u = D[L.U + it];
    S[L.r0 + it] = (TQ)(m.h * m.W[NX + k] * (u - uref(i, k)));
    S[L.lb + it] = (TQ)(m.ulb[k] - u);
    S[L.ub + it] = (TQ)(m.uub[k] - u);
  }
  if (mode & MODE_TRAJ) {  // expose the chunk like set_reference_trajectory's return value
    double* oy = st.yref + (size_t)b * N * NY;
    for (int it = tid; it < N * NY; it += 64) { const int i = it / NY, k = it - i * NY; oy[it] = k < NX ? xref(i, k) : m.uref[k - NX]; }
    if (tid < NX) st.yrefN[(size_t)b * NX + tid] = xref(N, tid);
  }
  TQ* gmu = st.mu + (size_t)b * 3 * nb;
  if (gp) {
    // alpha = Kx^-1 mu  (the OCP model evaluates k*(v_b) Kx^-1 p, src/gp/RGP.py:250-254)
    for (int i = tid; i < 3 * nb; i += 64) {
      const int d = i / nb, r = i % nb;
      TQ t = 0;
      for (int k = 0; k < nb; ++k) t += m.Kxinv[d * nb * nb + r * nb + k] * gmu[d * nb + k];
      S[L.alpha + i] = t;
      S[L.basis + i] = m.basis[i];
    }
  }
  __syncthreads();
  PF_STOP(PF_LOAD);
  // ---- 1. shooting
  shoot_states(m, D, S, L, gp);
  __syncthreads();
  PF_STOP(PF_SHOOT_X);
  shoot_sens(m, S, L);
  __syncthreads();
  PF_STOP(PF_SHOOT_S);   // shooting records (union region) are dead from here on
  if (tid < NX) S[L.dx + tid] = (TQ)(D[L.x0 + tid] - D[L.X + tid]);   // dx_0 = x_meas - X_0 (lbx = ubx = x_init)
  __syncthreads();
  // ---- 2. QP
  int status = 0;
  const int iters = solve_qp(m, S, L, &status PF_PASS);
  // ---- 3. full step (iterate accumulated in double)
  for (int i = tid; i < (N + 1) * NX; i += 64) { const double v = D[L.X + i] + (double)S[L.dx + i]; D[L.X + i] = v; gX[i] = v; }
  for (int i = tid; i < nv; i += 64) { const double v = D[L.U + i] + (double)S[L.z + i]; D[L.U + i] = v; gU[i] = v; }
  __syncthreads();
  // cost at the new iterate (get_cost)
  double cst = 0;
  int bad = 0;
  for (int it = tid; it < (N + 1) * NX; it += 64) {
    const int i = it / NX, k = it - i * NX;
    const double v = D[L.X + it], e = v - xref(i, k);
    cst += 0.5 * (i < N ? m.h * m.W[k] : m.We[k]) * e * e;
    if (!(v == v)) bad = 1;
  }
  for (int it = tid; it < nv; it += 64) {
    const double v = D[L.U + it], e = v - uref(it >> 2, it & 3);
    cst += 0.5 * m.h * m.W[NX + (it & 3)] * e * e;
    if (!(v == v)) bad = 1;
  }
  cst = wave_sum(cst);
  bad = wave_max(bad);
  if (bad) status = 1;
  if (tid == 0) { st.cost[b] = cst; st.status[b] = status; st.qp_iter[b] = iters; }
  if (tid < NU) st.w[(size_t)b * NU + tid] = D[L.U + tid];
  PF_START();
  if (!(mode & MODE_POST)) return;
  // ---- 4. post: nominal prediction, cursor, drag estimate, RGP regress, statistics
  double* vbad = D + L.x0 + NX;   // [v_body(3), a_drag(3)]
  if (tid == 0) {
    double x[NX], u[NU], xp[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = D[L.x0 + k];
#pragma unroll
    for (int k = 0; k < NU; ++k) u[k] = D[L.U + k];
    rk4_nominal(m, x, u, m.dt_pred, xp);
    // compute_a_drag (src/utils/utils.py:934-950) against the previous step's prediction
    double xq[NX];
    const bool hp = st.has_prev[b] != 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) xq[k] = hp ? st.xpp[(size_t)b * NX + k] : x[k];
    double R[9], Rq[9];
    rotmat(x + 3, R);
    rotmat(xq + 3, Rq);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double vb = R[i] * x[7] + R[3 + i] * x[8] + R[6 + i] * x[9];
      const double vp = Rq[i] * xq[7] + Rq[3 + i] * xq[8] + Rq[6 + i] * xq[9];
      vbad[i] = vb;
      vbad[3 + i] = (vb - vp) / m.dt_pred;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) { st.xpred[(size_t)b * NX + k] = xp[k]; st.xpp[(size_t)b * NX + k] = xp[k]; }
    st.has_prev[b] = 1;
    st.idx[b] = idx + 1;
    // tracking statistic against the first row of the reference chunk
    double ep = 0, ev = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double a = x[k] - xref(0, k), c = x[7 + k] - xref(0, 7 + k);
      ep += a * a; ev += c * c;
    }
    double* gs = st.stats + (size_t)b * 4;
    gs[0] += ep; gs[1] += ev; gs[2] += 1; gs[3] = tmax(gs[3], ep);
  }
  __syncthreads();
  if (gp) rgp_regress(m, S, L, gmu, st.C + (size_t)b * 3 * nb * nb, vbad, vbad + 3);
#ifdef MPCQ_PROFILE
  PF_STOP(PF_POST);
  pf.acc[PF_TOTAL] = __builtin_readcyclecounter() - t_begin;
  if (tid == 0 && st.prof)
    for (int k = 0; k < PF_N; ++k) st.prof[(size_t)b * PF_N + k] = pf.acc[k];
#endif
}

// ------------------------------------------------------------------ small explicit-path kernels
template <typename TQ>
__global__ void predict_kernel(const DevModel<TQ> m, const double* x, const double* u, double dt, double* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double xi[NX], ui[NU], xo[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k) xi[k] = x[(size_t)b * NX + k];
#pragma unroll
  for (int k = 0; k < NU; ++k) ui[k] = u[(size_t)b * NU + k];
  rk4_nominal(m, xi, ui, dt, xo);
#pragma unroll
  for (int k = 0; k < NX; ++k) out[(size_t)b * NX + k] = xo[k];
}

template <typename TQ>
__global__ void regress_kernel(const DevModel<TQ> m, const DevState<TQ> st, const double* vb, const double* ad) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const Lds L = lds_layout(m.N, m.nb);
  TQ* S = reinterpret_cast<TQ*>(smem_raw + L.dbytes);
  const int b = blockIdx.x;
  rgp_regress(m, S, L, st.mu + (size_t)b * 3 * m.nb, st.C + (size_t)b * 3 * m.nb * m.nb, vb + (size_t)b * 3, ad + (size_t)b * 3);
}

// closed-loop plant: n_sub RK4 substeps of the drag plant from the engine's plant state with input w
template <typename TQ>
__global__ void plant_kernel(const DevModel<TQ> m, double* xs, const double* w, int n_sub, double sim_dt, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double x[NX], u[NU];
#pragma unroll
  for (int k = 0; k < NX; ++k) x[k] = xs[(size_t)b * NX + k];
#pragma unroll
  for (int k = 0; k < NU; ++k) u[k] = w[(size_t)b * NU + k];
  for (int s = 0; s < n_sub; ++s) plant_rk4(m, x, u, sim_dt);
#pragma unroll
  for (int k = 0; k < NX; ++k) xs[(size_t)b * NX + k] = x[k];
}

// reduce per-instance statistics to 5 numbers (sum, sum, sum, max, #failed)
__global__ void stats_kernel(const double* stats, const int* status, int B, double* out5) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double (*sh)[256] = reinterpret_cast<double (*)[256]>(smem_raw);  // [5][256]
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    a0 += stats[(size_t)b * 4]; a1 += stats[(size_t)b * 4 + 1]; a2 += stats[(size_t)b * 4 + 2];
    const double mx = stats[(size_t)b * 4 + 3];
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
