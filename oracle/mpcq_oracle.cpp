// oracle/mpcq_oracle.cpp
//
// TEST INFRASTRUCTURE — NOT THE PRODUCT.
//
// CPU fp64 restatement of the per-timestep MPC+RGP control loop of
// smidmatej/mpc_quad_ros (reference paths below are relative to the reference
// checkout).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library; the shipped path (mpc_quad_ros_amd/, libmpcq.so)
// never links, imports or calls it.
//
// Parity status: PINNED.  tests/test_oracle_golden.py replays the reference's
// own logged acados+HPIPM runs (tests/golden/*.npz, produced from
// outputs/*/data/*.pkl by tests/golden/make_golden.py) through this code and
// checks w_odom / cost_solution / x_pred_odom / rgp_mu_g_t / rgp_C_g_t /
// v_body / a_drag / plant states.  The RGP half is additionally pinned against
// vectors produced by importing the reference's src/gp/RGP.py.
//
// The MPC arithmetic of the reference lives in third-party code that is not in
// the reference tree (acados + HPIPM + BLASFEO + casadi codegen, unpinned,
// reached through acados_template.AcadosOcpSolver at src/quad_opt.py:25,156).
// What is restated here is the published SQP-RTI algorithm with the option set
// dumped in src/_acados_ocp.json:2082-2151 (ERK 4 stages / 1 step, Gauss-Newton,
// LINEAR_LS cost scaled by the shooting interval, full step, no shift), with
// the QP solved to its unique optimum instead of following HPIPM's iterates.
//
// Every function cites the reference lines it follows.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int NX = 13, NU = 4, NY = 17;

struct Model {
  int N = 0;       // shooting intervals (n_nodes, src/quad_opt.py:39)
  double T = 1.0;  // horizon (t_horizon, src/quad_opt.py:40)
  int nb = 0;      // RGP basis points per axis; 0 = no GP in the model
  double dt_pred = 0.01;  // nominal prediction step (ODOMETRY_DT node / optimization_dt sim)
  int skip = 1;           // control_freq_factor, src/mpc_controller_node.py:222
  // quad constants, src/quad.py:41-93,385-417
  double mass, J[3], tmax, xf[4], yf[4], zl[4], g;
  double rotor_drag[3], aero_drag;  // plant only, src/quad.py:79-89
  double W[NY], We[NX];             // src/quad_opt.py:122-130
  double ulb[NU], uub[NU], uref[NU];  // src/quad_opt.py:142-144,304
  std::vector<double> Xb[3];          // basis vectors
  double L[3], sf[3], sn[3];          // theta = [L, sigma_f, sigma_n], src/gp/RGP.py:131-136
  std::vector<double> Kx[3], Kxinv[3];
  double qp_tol = 1e-12;
};

// ---------------------------------------------------------------- quaternion helpers
// q_to_rot_mat, src/utils/utils.py:325-340 (unnormalised formula, reproduced as written)
inline void rotmat(const double* q, double R[9]) {
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz);     R[2] = 2 * (qx * qz + qw * qy);
  R[3] = 2 * (qx * qy + qw * qz);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
  R[6] = 2 * (qx * qz - qw * qy);     R[7] = 2 * (qy * qz + qw * qx);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}
// dR/dq_i (derived), i = w,x,y,z
inline void drotmat(const double* q, double dR[4][9]) {
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const double a[4][9] = {
      {0, -qz, qy, qz, 0, -qx, -qy, qx, 0},
      {0, qy, qz, qy, -2 * qx, -qw, qz, qw, -2 * qx},
      {-2 * qy, qx, qw, qx, 0, qz, -qw, qz, -2 * qy},
      {-2 * qz, -qw, qx, qw, -2 * qz, qy, qx, qy, 0}};
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 9; ++k) dR[i][k] = 2 * a[i][k];
}
// v_dot_q(v, q) = R(q) v, src/utils/utils.py:317-322
inline void rot(const double R[9], const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
// v_dot_q(v, quaternion_inverse(q)) = R(conj q) v = R(q)^T v, src/utils/utils.py:434-440
inline void rot_conj(const double* q, const double* v, double* o) {
  double qc[4] = {q[0], -q[1], -q[2], -q[3]}, R[9];
  rotmat(qc, R);
  rot(R, v, o);
}

// RBF.__call__, src/gp/RGP.py:40-58: sigma_f**2 * exp(-1/2*(x1-x2) * inv(L*L) * (x1-x2))
inline double rbf(double x1, double x2, double L, double sf) {
  const double d = x1 - x2;
  const double invLL = 1.0 / (L * L);
  return sf * sf * std::exp(((-0.5 * d) * invLL) * d);
}

// ---------------------------------------------------------------- model
// setup_casadi_model, src/quad_opt.py:164-262.  alpha = K_x^-1 mu per axis (the casadi
// graph evaluates k*(v_b) K_x^-1 p, src/gp/RGP.py:250-254); alpha == nullptr -> nominal model.
// Jac (optional) is d f / d [x,u], row-major 13x17, analytic.
void model_f(const Model& m, const double* x, const double* u, const double* alpha, double* f,
             double* Jac) {
  const double* q = x + 3;
  const double* v = x + 7;
  const double* r = x + 10;
  double R[9];
  rotmat(q, R);
  // f_p, src/quad_opt.py:187
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  // f_q = 1/2 skew_symmetric(r) q, src/quad_opt.py:190, src/utils/utils.py:408-412
  f[3] = 0.5 * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = 0.5 * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = 0.5 * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = 0.5 * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  // f_v, src/quad_opt.py:193-196
  const double aT = (u[0] * m.tmax + u[1] * m.tmax + u[2] * m.tmax + u[3] * m.tmax) / m.mass;
  f[7] = R[2] * aT;
  f[8] = R[5] * aT;
  f[9] = R[8] * aT - m.g;
  // f_r, src/quad_opt.py:203-206
  double ty = 0, tx = 0, tz = 0;
  for (int j = 0; j < 4; ++j) {
    ty += u[j] * m.tmax * m.yf[j];
    tx += u[j] * m.tmax * m.xf[j];
    tz += u[j] * m.tmax * m.zl[j];
  }
  f[10] = (ty + (m.J[1] - m.J[2]) * r[1] * r[2]) / m.J[0];
  f[11] = (-tx + (m.J[2] - m.J[0]) * r[2] * r[0]) / m.J[1];
  f[12] = (tz + (m.J[0] - m.J[1]) * r[0] * r[1]) / m.J[2];

  // RGP augmentation, src/quad_opt.py:211-251
  double vb[3] = {0, 0, 0}, mg[3] = {0, 0, 0}, mgp[3] = {0, 0, 0};
  const bool gp = alpha != nullptr && m.nb > 0;
  if (gp) {
    for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];  // R^T v
    for (int d = 0; d < 3; ++d) {
      const double invLL = 1.0 / (m.L[d] * m.L[d]);
      for (int j = 0; j < m.nb; ++j) {
        const double k = rbf(vb[d], m.Xb[d][j], m.L[d], m.sf[d]);
        mg[d] += alpha[d * m.nb + j] * k;
        mgp[d] += alpha[d * m.nb + j] * k * (m.Xb[d][j] - vb[d]) * invLL;
      }
    }
    double a[3];
    rot(R, mg, a);
    f[7] += a[0]; f[8] += a[1]; f[9] += a[2];
  }
  if (!Jac) return;
  std::fill(Jac, Jac + NX * NY, 0.0);
  auto Jx = [&](int i, int j) -> double& { return Jac[i * NY + j]; };
  // d pdot / d v
  for (int i = 0; i < 3; ++i) Jx(i, 7 + i) = 1.0;
  // d qdot / d q = 1/2 Omega(r)
  const double Om[16] = {0, -r[0], -r[1], -r[2], r[0], 0, r[2], -r[1],
                         r[1], -r[2], 0, r[0], r[2], r[1], -r[0], 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Jx(3 + i, 3 + j) = 0.5 * Om[4 * i + j];
  // d qdot / d r
  const double Qr[12] = {-q[1], -q[2], -q[3], q[0], -q[3], q[2], q[3], q[0], -q[1], -q[2], q[1], q[0]};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) Jx(3 + i, 10 + j) = 0.5 * Qr[3 * i + j];
  // d vdot / d q, d vdot / d v, d vdot / d u
  double dR[4][9];
  drotmat(q, dR);
  for (int i = 0; i < 4; ++i) {
    for (int row = 0; row < 3; ++row) {
      double val = aT * dR[i][3 * row + 2];
      if (gp) {
        // (dR/dq_i) m + R diag(m') (dR/dq_i)^T v
        double t = 0;
        for (int c = 0; c < 3; ++c) t += dR[i][3 * row + c] * mg[c];
        for (int c = 0; c < 3; ++c) {
          double dvb = dR[i][c] * v[0] + dR[i][3 + c] * v[1] + dR[i][6 + c] * v[2];
          t += R[3 * row + c] * mgp[c] * dvb;
        }
        val += t;
      }
      Jx(7 + row, 3 + i) = val;
    }
  }
  if (gp) {
    for (int row = 0; row < 3; ++row)
      for (int col = 0; col < 3; ++col) {
        double t = 0;
        for (int c = 0; c < 3; ++c) t += R[3 * row + c] * mgp[c] * R[3 * col + c];
        Jx(7 + row, 7 + col) = t;
      }
  }
  for (int j = 0; j < 4; ++j)
    for (int row = 0; row < 3; ++row) Jx(7 + row, 13 + j) = R[3 * row + 2] * m.tmax / m.mass;
  // d rdot / d r, d rdot / d u
  Jx(10, 11) = (m.J[1] - m.J[2]) * r[2] / m.J[0]; Jx(10, 12) = (m.J[1] - m.J[2]) * r[1] / m.J[0];
  Jx(11, 10) = (m.J[2] - m.J[0]) * r[2] / m.J[1]; Jx(11, 12) = (m.J[2] - m.J[0]) * r[0] / m.J[1];
  Jx(12, 10) = (m.J[0] - m.J[1]) * r[1] / m.J[2]; Jx(12, 11) = (m.J[0] - m.J[1]) * r[0] / m.J[2];
  for (int j = 0; j < 4; ++j) {
    Jx(10, 13 + j) = m.tmax * m.yf[j] / m.J[0];
    Jx(11, 13 + j) = -m.tmax * m.xf[j] / m.J[1];
    Jx(12, 13 + j) = m.tmax * m.zl[j] / m.J[2];
  }
}

// quad_optimizer.discrete_dynamics, src/quad_opt.py:353-377 (one RK4 step, same expression order)
void rk4(const Model& m, const double* x, const double* u, const double* alpha, double dt, double* xo) {
  double k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
  model_f(m, x, u, alpha, k1, nullptr);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k1[i];
  model_f(m, xt, u, alpha, k2, nullptr);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k2[i];
  model_f(m, xt, u, alpha, k3, nullptr);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
  model_f(m, xt, u, alpha, k4, nullptr);
  for (int i = 0; i < NX; ++i) xo[i] = x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

// acados ERK (4 stages, 1 step, src/_acados_ocp.json:2087,2126-2149) with forward
// sensitivities: phi = Phi(x,u), AB = [dPhi/dx | dPhi/du] (13x17 row-major).
void rk4_sens(const Model& m, const double* x, const double* u, const double* alpha, double h,
              double* phi, double* AB) {
  double k[4][NX], S[4][NX * NY], Jc[NX * NY], xt[NX], Z[NX * NY];
  const double a[4] = {0, 0.5, 0.5, 1.0};
  for (int s = 0; s < 4; ++s) {
    for (int i = 0; i < NX; ++i) xt[i] = x[i] + (s ? h * a[s] * k[s - 1][i] : 0.0);
    model_f(m, xt, u, alpha, k[s], Jc);
    // Z = [I|0] + h a_s S_{s-1}
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NY; ++j)
        Z[i * NY + j] = (i == j ? 1.0 : 0.0) + (s ? h * a[s] * S[s - 1][i * NY + j] : 0.0);
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NY; ++j) {
        double t = (j >= NX) ? Jc[i * NY + j] : 0.0;
        for (int l = 0; l < NX; ++l) t += Jc[i * NY + l] * Z[l * NY + j];
        S[s][i * NY + j] = t;
      }
  }
  for (int i = 0; i < NX; ++i) phi[i] = x[i] + h / 6 * (k[0][i] + 2 * k[1][i] + 2 * k[2][i] + k[3][i]);
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NY; ++j)
      AB[i * NY + j] = (i == j ? 1.0 : 0.0) +
                       h / 6 * (S[0][i * NY + j] + 2 * S[1][i * NY + j] + 2 * S[2][i * NY + j] + S[3][i * NY + j]);
}

// ---------------------------------------------------------------- dense linear algebra
bool cholesky(std::vector<double>& A, int n) {  // in-place lower
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return true;
}
void chol_solve(const std::vector<double>& Lm, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= Lm[i * n + k] * b[k];
    b[i] = s / Lm[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= Lm[k * n + i] * b[k];
    b[i] = s / Lm[i * n + i];
  }
}
// inverse by Gauss-Jordan with partial pivoting (np.linalg.inv equivalent, src/gp/RGP.py:157)
bool invert(const std::vector<double>& A, int n, std::vector<double>& Ai) {
  std::vector<double> M(A);
  Ai.assign(n * n, 0.0);
  for (int i = 0; i < n; ++i) Ai[i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int i = c + 1; i < n; ++i)
      if (std::fabs(M[i * n + c]) > std::fabs(M[p * n + c])) p = i;
    if (M[p * n + c] == 0.0) return false;
    if (p != c)
      for (int j = 0; j < n; ++j) {
        std::swap(M[p * n + j], M[c * n + j]);
        std::swap(Ai[p * n + j], Ai[c * n + j]);
      }
    const double d = 1.0 / M[c * n + c];
    for (int j = 0; j < n; ++j) { M[c * n + j] *= d; Ai[c * n + j] *= d; }
    for (int i = 0; i < n; ++i) {
      if (i == c) continue;
      const double fct = M[i * n + c];
      if (fct == 0.0) continue;
      for (int j = 0; j < n; ++j) { M[i * n + j] -= fct * M[c * n + j]; Ai[i * n + j] -= fct * Ai[c * n + j]; }
    }
  }
  return true;
}

// ---------------------------------------------------------------- box QP
// min 1/2 z'Hz + g'z  s.t. lb <= z <= ub, H SPD (strictly convex: unique optimum, so the
// answer does not depend on the method; the reference reaches it with HPIPM's dense IPM,
// src/_acados_ocp.json:2085,2104-2116).  Mehrotra predictor-corrector to mu ~ 1e-13, then an
// active-set polish that solves the KKT system of the identified active set exactly.
// returns iterations (>0) or negative on failure.
// Scratch of one solve, kept per thread and reused from call to call (the batched driver runs thousands of solves per
// thread: allocating ~0.5 MB of std::vectors in every one of them serialises the OpenMP threads on the heap).
struct Work {
  std::vector<double> sl, su, ll, lu, rd, dz, dza, M, rhs, dll, dlu, zz, grad, Hf, bf;
  std::vector<int> act, fr;
  std::vector<double> alpha, AB, c, d, G, H, g, lb, ub, z, dx, chunk;
};
Work& work() { static thread_local Work w; return w; }

int box_qp(int n, const std::vector<double>& H, const std::vector<double>& g, const std::vector<double>& lb,
           const std::vector<double>& ub, std::vector<double>& z, double tol, double* kkt_out) {
  Work& ws = work();
  std::vector<double>&sl = ws.sl, &su = ws.su, &ll = ws.ll, &lu = ws.lu, &rd = ws.rd, &dz = ws.dz, &dza = ws.dza, &M = ws.M, &rhs = ws.rhs;
  std::vector<double>&dll = ws.dll, &dlu = ws.dlu;
  for (std::vector<double>* v : {&sl, &su, &ll, &lu, &rd, &dz, &dza, &rhs, &dll, &dlu}) v->assign(n, 0.0);
  M.assign((size_t)n * n, 0.0);
  double gmax = 1.0;
  for (int i = 0; i < n; ++i) gmax = std::max(gmax, std::fabs(g[i]));
  for (int i = 0; i < n; ++i) {
    const double w = ub[i] - lb[i];
    double z0 = std::min(std::max(0.0, lb[i] + 0.1 * w), ub[i] - 0.1 * w);
    z[i] = z0;
    sl[i] = z0 - lb[i];
    su[i] = ub[i] - z0;
    ll[i] = gmax / sl[i] * 0.1;
    lu[i] = gmax / su[i] * 0.1;
  }
  int it = 0;
  const int maxit = 80;
  for (; it < maxit; ++it) {
    double mu = 0, rdmax = 0;
    for (int i = 0; i < n; ++i) {
      double t = g[i] - ll[i] + lu[i];
      for (int j = 0; j < n; ++j) t += H[i * n + j] * z[j];
      rd[i] = t;
      rdmax = std::max(rdmax, std::fabs(t));
      mu += sl[i] * ll[i] + su[i] * lu[i];
    }
    mu /= 2 * n;
    if (rdmax < tol * gmax && mu < tol) break;
    M = H;
    for (int i = 0; i < n; ++i) M[i * n + i] += ll[i] / sl[i] + lu[i] / su[i];
    if (!cholesky(M, n)) return -1;
    // predictor: sigma = 0
    for (int i = 0; i < n; ++i) rhs[i] = -rd[i] - ll[i] + lu[i];
    dza = rhs;
    chol_solve(M, n, dza.data());
    double aff = 1.0;
    for (int i = 0; i < n; ++i) {
      const double dl = -ll[i] - ll[i] / sl[i] * dza[i];
      const double du = -lu[i] + lu[i] / su[i] * dza[i];
      if (dza[i] < 0) aff = std::min(aff, -sl[i] / dza[i]);
      if (dza[i] > 0) aff = std::min(aff, su[i] / dza[i]);
      if (dl < 0) aff = std::min(aff, -ll[i] / dl);
      if (du < 0) aff = std::min(aff, -lu[i] / du);
      dll[i] = dl;
      dlu[i] = du;
    }
    double mua = 0;
    for (int i = 0; i < n; ++i)
      mua += (sl[i] + aff * dza[i]) * (ll[i] + aff * dll[i]) + (su[i] - aff * dza[i]) * (lu[i] + aff * dlu[i]);
    mua /= 2 * n;
    const double sigma = std::pow(mua / mu, 3.0);
    // corrector
    for (int i = 0; i < n; ++i) {
      const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i];
      const double rcu = -su[i] * lu[i] + sigma * mu - (-dza[i]) * dlu[i];
      rhs[i] = -rd[i] + rcl / sl[i] - rcu / su[i];
    }
    dz = rhs;
    chol_solve(M, n, dz.data());
    double ap = 1.0, ad = 1.0;
    for (int i = 0; i < n; ++i) {
      const double rcl = -sl[i] * ll[i] + sigma * mu - dza[i] * dll[i];
      const double rcu = -su[i] * lu[i] + sigma * mu - (-dza[i]) * dlu[i];
      dll[i] = (rcl - ll[i] * dz[i]) / sl[i];
      dlu[i] = (rcu + lu[i] * dz[i]) / su[i];
      if (dz[i] < 0) ap = std::min(ap, -sl[i] / dz[i]);
      if (dz[i] > 0) ap = std::min(ap, su[i] / dz[i]);
      if (dll[i] < 0) ad = std::min(ad, -ll[i] / dll[i]);
      if (dlu[i] < 0) ad = std::min(ad, -lu[i] / dlu[i]);
    }
    const double tau = std::max(0.995, 1.0 - mu);
    ap = std::min(1.0, tau * ap);
    ad = std::min(1.0, tau * ad);
    for (int i = 0; i < n; ++i) {
      z[i] += ap * dz[i];
      sl[i] += ap * dz[i];
      su[i] -= ap * dz[i];
      ll[i] += ad * dll[i];
      lu[i] += ad * dlu[i];
    }
  }
  // ---- active-set polish
  std::vector<int>& act = ws.act;  // -1 lower, +1 upper, 0 free
  act.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    if (ll[i] > sl[i]) act[i] = -1;
    else if (lu[i] > su[i]) act[i] = 1;
  }
  std::vector<double>&zz = ws.zz, &grad = ws.grad;
  zz.assign(n, 0.0); grad.assign(n, 0.0);
  double kkt = 1e300;
  for (int pass = 0; pass < 4 * n + 10; ++pass) {
    std::vector<int>& fr = ws.fr;
    fr.clear();
    for (int i = 0; i < n; ++i) {
      if (act[i] == 0) fr.push_back(i);
      else zz[i] = act[i] < 0 ? lb[i] : ub[i];
    }
    const int nf = (int)fr.size();
    if (nf) {
      std::vector<double>&Hf = ws.Hf, &bf = ws.bf;
      Hf.assign((size_t)nf * nf, 0.0); bf.assign(nf, 0.0);
      for (int a = 0; a < nf; ++a) {
        double t = -g[fr[a]];
        for (int j = 0; j < n; ++j)
          if (act[j] != 0) t -= H[fr[a] * n + j] * zz[j];
        bf[a] = t;
        for (int b = 0; b < nf; ++b) Hf[a * nf + b] = H[fr[a] * n + fr[b]];
      }
      if (!cholesky(Hf, nf)) return -2;
      chol_solve(Hf, nf, bf.data());
      for (int a = 0; a < nf; ++a) zz[fr[a]] = bf[a];
    }
    for (int i = 0; i < n; ++i) {
      double t = g[i];
      for (int j = 0; j < n; ++j) t += H[i * n + j] * zz[j];
      grad[i] = t;
    }
    // worst violation
    int worst = -1;
    double wv = 1e-13;
    for (int i = 0; i < n; ++i) {
      double viol = 0;
      if (act[i] == 0) viol = std::max(lb[i] - zz[i], zz[i] - ub[i]);  // primal infeasible
      else if (act[i] < 0) viol = -grad[i];                             // multiplier lambda_l = grad >= 0
      else viol = grad[i];                                              // lambda_u = -grad >= 0
      if (viol > wv) { wv = viol; worst = i; }
    }
    if (worst < 0) {
      kkt = 0;
      for (int i = 0; i < n; ++i)
        if (act[i] == 0) kkt = std::max(kkt, std::fabs(grad[i]));
      z = zz;
      if (kkt_out) *kkt_out = kkt;
      return it + 1;
    }
    if (act[worst] == 0) act[worst] = (zz[worst] < lb[worst]) ? -1 : 1;
    else act[worst] = 0;
  }
  // polish failed to settle (degenerate cycling): keep the IPM point (KKT <= tol)
  if (kkt_out) *kkt_out = -1;
  return it + 1;
}

// ---------------------------------------------------------------- one SQP-RTI call
// acados solve() as driven by quad_optimizer.run_optimization, src/quad_opt.py:321-350.
// X ((N+1)x13), U (Nx4): persisted iterate, updated in place (full step, no shift).
// yref (Nx17), yrefN (13): set_reference_trajectory, src/quad_opt.py:295-317.
// mu (3*nb) or nullptr: stage parameters p (same on all stages, src/quad_opt.py:402-404).
struct RtiOut { double cost; int status; int qp_iter; double kkt; };

RtiOut rti_solve(const Model& m, double* X, double* U, const double* x0, const double* yref,
                 const double* yrefN, const double* mu) {
  const int N = m.N, nv = NU * N;
  const double h = m.T / N;  // optimization_dt, src/quad_opt.py:43
  RtiOut out{0, 0, 0, 0};
  Work& ws = work();
  std::vector<double>& alpha = ws.alpha;
  const double* al = nullptr;
  if (m.nb > 0 && mu) {
    alpha.assign(3 * m.nb, 0.0);
    for (int d = 0; d < 3; ++d)
      for (int i = 0; i < m.nb; ++i) {
        double t = 0;
        for (int j = 0; j < m.nb; ++j) t += m.Kxinv[d][i * m.nb + j] * mu[d * m.nb + j];
        alpha[d * m.nb + i] = t;
      }
    al = alpha.data();
  }
  // 1. shooting: phi_i, [A_i|B_i], gap c_i = phi_i - X_{i+1}
  std::vector<double>&AB = ws.AB, &c = ws.c;
  AB.assign((size_t)N * NX * NY, 0.0); c.assign((size_t)N * NX, 0.0);
  for (int i = 0; i < N; ++i) {
    double phi[NX];
    rk4_sens(m, X + i * NX, U + i * NU, al, h, phi, &AB[i * NX * NY]);
    for (int k = 0; k < NX; ++k) c[i * NX + k] = phi[k] - X[(i + 1) * NX + k];
  }
  // 2. condensing.  dx_i = d_i + sum_{j<i} G[i][j] du_j ; d_0 = x0 - X_0 (lbx=ubx=x_init, :328-329)
  std::vector<double>&d = ws.d, &G = ws.G;
  d.assign((size_t)(N + 1) * NX, 0.0); G.assign((size_t)(N + 1) * N * NX * NU, 0.0);
  auto Gb = [&](int i, int j) { return &G[((size_t)i * N + j) * NX * NU]; };
  for (int k = 0; k < NX; ++k) d[k] = x0[k] - X[k];
  for (int i = 0; i < N; ++i) {
    const double* A = &AB[i * NX * NY];
    for (int r = 0; r < NX; ++r) {
      double t = c[i * NX + r];
      for (int k = 0; k < NX; ++k) t += A[r * NY + k] * d[i * NX + k];
      d[(i + 1) * NX + r] = t;
    }
    for (int j = 0; j < i; ++j) {
      const double* Gi = Gb(i, j);
      double* Go = Gb(i + 1, j);
      for (int r = 0; r < NX; ++r)
        for (int cc = 0; cc < NU; ++cc) {
          double t = 0;
          for (int k = 0; k < NX; ++k) t += A[r * NY + k] * Gi[k * NU + cc];
          Go[r * NU + cc] = t;
        }
    }
    double* Go = Gb(i + 1, i);
    for (int r = 0; r < NX; ++r)
      for (int cc = 0; cc < NU; ++cc) Go[r * NU + cc] = A[r * NY + NX + cc];
  }
  // cost: stage i<N weight h*W (acados scales LS stage costs by the interval; terminal unscaled)
  std::vector<double>&H = ws.H, &g = ws.g, &lb = ws.lb, &ub = ws.ub, &z = ws.z;
  H.assign((size_t)nv * nv, 0.0); g.assign(nv, 0.0); lb.assign(nv, 0.0); ub.assign(nv, 0.0); z.assign(nv, 0.0);
  for (int i = 1; i <= N; ++i) {
    double qd[NX], e[NX];
    for (int k = 0; k < NX; ++k) {
      qd[k] = (i < N) ? h * m.W[k] : m.We[k];
      const double ref = (i < N) ? yref[i * NY + k] : yrefN[k];
      e[k] = qd[k] * (d[i * NX + k] + X[i * NX + k] - ref);
    }
    for (int j = 0; j < i; ++j) {
      const double* Gj = Gb(i, j);
      for (int a = 0; a < NU; ++a) {
        double t = 0;
        for (int k = 0; k < NX; ++k) t += Gj[k * NU + a] * e[k];
        g[j * NU + a] += t;
      }
      for (int l = 0; l <= j; ++l) {
        const double* Gl = Gb(i, l);
        for (int a = 0; a < NU; ++a)
          for (int b = 0; b < NU; ++b) {
            double t = 0;
            for (int k = 0; k < NX; ++k) t += Gj[k * NU + a] * qd[k] * Gl[k * NU + b];
            H[(j * NU + a) * nv + l * NU + b] += t;
          }
      }
    }
  }
  for (int j = 0; j < N; ++j)
    for (int a = 0; a < NU; ++a) {
      const int ia = j * NU + a;
      H[ia * nv + ia] += h * m.W[NX + a];
      g[ia] += h * m.W[NX + a] * (U[ia] - yref[j * NY + NX + a]);
      lb[ia] = m.ulb[a] - U[ia];
      ub[ia] = m.uub[a] - U[ia];
    }
  for (int i = 0; i < nv; ++i)
    for (int j = i + 1; j < nv; ++j) H[i * nv + j] = H[j * nv + i];
  // 3. QP
  double kkt = 0;
  const int it = box_qp(nv, H, g, lb, ub, z, m.qp_tol, &kkt);
  out.qp_iter = it;
  out.kkt = kkt;
  if (it < 0) { out.status = 4; return out; }  // ACADOS_QP_FAILURE
  // 4. expand + full step (nlp_solver_step_length 1.0, src/_acados_ocp.json:2094)
  std::vector<double>& dx = ws.dx;
  dx.assign((size_t)(N + 1) * NX, 0.0);
  for (int k = 0; k < NX; ++k) dx[k] = d[k];
  for (int i = 0; i < N; ++i) {
    const double* A = &AB[i * NX * NY];
    for (int r = 0; r < NX; ++r) {
      double t = c[i * NX + r];
      for (int k = 0; k < NX; ++k) t += A[r * NY + k] * dx[i * NX + k];
      for (int k = 0; k < NU; ++k) t += A[r * NY + NX + k] * z[i * NU + k];
      dx[(i + 1) * NX + r] = t;
    }
  }
  for (int i = 0; i <= N; ++i)
    for (int k = 0; k < NX; ++k) X[i * NX + k] += dx[i * NX + k];
  for (int i = 0; i < nv; ++i) U[i] += z[i];
  // 5. get_cost(): objective at the new iterate, src/quad_opt.py:350
  double cost = 0;
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < NX; ++k) { const double e = X[i * NX + k] - yref[i * NY + k]; cost += 0.5 * h * m.W[k] * e * e; }
    for (int k = 0; k < NU; ++k) { const double e = U[i * NU + k] - yref[i * NY + NX + k]; cost += 0.5 * h * m.W[NX + k] * e * e; }
  }
  for (int k = 0; k < NX; ++k) { const double e = X[N * NX + k] - yrefN[k]; cost += 0.5 * m.We[k] * e * e; }
  out.cost = cost;
  for (int i = 0; i < (N + 1) * NX; ++i)
    if (!std::isfinite(X[i])) out.status = 1;  // ACADOS_NAN_DETECTED
  return out;
}

// ---------------------------------------------------------------- RGP
// RGP.__init__, src/gp/RGP.py:126-157: mu0 = 0 (y_), C0 = K(X,X) + sigma_n^2 I, K_x = C0, K_x_inv.
void rgp_setup(Model& m) {
  for (int d = 0; d < 3; ++d) {
    const int n = m.nb;
    m.Kx[d].assign(n * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        m.Kx[d][i * n + j] = rbf(m.Xb[d][i], m.Xb[d][j], m.L[d], m.sf[d]) + (i == j ? m.sn[d] * m.sn[d] : 0.0);
    invert(m.Kx[d], n, m.Kxinv[d]);
  }
}

// RGP.regress (src/gp/RGP.py:303-330) through RGP.predict(cov=True, return_Jt=True)
// (src/gp/RGP.py:199-208) for ONE new point (s, y) on axis d.  Operation order follows numpy's.
void rgp_regress_axis(const Model& m, int d, double s, double y, double* mu, double* C) {
  const int n = m.nb;
  static thread_local std::vector<double> ks, Jt, JC, G, GJ, Cn;
  ks.assign(n, 0.0); Jt.assign(n, 0.0); JC.assign(n, 0.0); G.assign(n, 0.0); GJ.assign((size_t)n * n, 0.0); Cn.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) ks[j] = rbf(s, m.Xb[d][j], m.L[d], m.sf[d]);
  for (int j = 0; j < n; ++j) {
    double t = 0;
    for (int i = 0; i < n; ++i) t += ks[i] * m.Kxinv[d][i * n + j];
    Jt[j] = t;
  }
  double mup = 0, Jk = 0;
  for (int j = 0; j < n; ++j) { mup += Jt[j] * mu[j]; Jk += Jt[j] * ks[j]; }
  const double Bv = rbf(s, s, m.L[d], m.sf[d]) - Jk;  // K(Xt,Xt) - Jt K(X,Xt): no noise term
  for (int j = 0; j < n; ++j) {
    double t = 0;
    for (int i = 0; i < n; ++i) t += Jt[i] * C[i * n + j];
    JC[j] = t;
  }
  double JCJ = 0;
  for (int j = 0; j < n; ++j) JCJ += JC[j] * Jt[j];
  const double Cp = Bv + JCJ;
  const double inv = 1.0 / (Cp + m.sn[d] * m.sn[d]);
  for (int i = 0; i < n; ++i) {
    double t = 0;
    for (int j = 0; j < n; ++j) t += C[i * n + j] * Jt[j];
    G[i] = t * inv;
  }
  for (int i = 0; i < n; ++i) mu[i] = mu[i] + G[i] * (y - mup);
  // C - (G Jt) C   (G_tilde_t.dot(Jt).dot(C_g_t_minus_1); not symmetrised)
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) GJ[i * n + j] = G[i] * Jt[j];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double t = 0;
      for (int k = 0; k < n; ++k) t += GJ[i * n + k] * C[k * n + j];
      Cn[i * n + j] = C[i * n + j] - t;
    }
  std::copy(Cn.begin(), Cn.end(), C);
}

// compute_a_drag, src/utils/utils.py:934-950
void compute_a_drag(const double* x, const double* xpm1, double dt, double* vb, double* ad) {
  double vp[3];
  rot_conj(x + 3, x + 7, vb);
  rot_conj(xpm1 + 3, xpm1 + 7, vp);
  for (int i = 0; i < 3; ++i) ad[i] = (vb[i] - vp[i]) / dt;
}

// get_reference_chunk, src/utils/utils.py:897-931 (numpy slice semantics restated with ints)
void reference_chunk(const double* traj, int len, int idx, int N, int skip, double* out) {
  const long left = (long)len - idx;
  const double* last = traj + (size_t)(len - 1) * NX;
  int have = 0;
  if (left > (long)N * skip) {
    have = N;
  } else if (left > skip - 1) {
    // traj[idx : idx + left*skip : skip] clipped at len -> ceil(left/skip) rows
    have = (int)((left + skip - 1) / skip);
    if (have > N) have = N;
  }
  for (int j = 0; j < have; ++j) std::memcpy(out + j * NX, traj + (size_t)(idx + (long)j * skip) * NX, NX * sizeof(double));
  for (int j = have; j < N; ++j) std::memcpy(out + j * NX, last, NX * sizeof(double));
}

// ---------------------------------------------------------------- plant (harness)
// Quadrotor3D.f_nominal with drag=True, payload=False: src/quad.py:256-381
void plant_f(const Model& m, const double* x, const double* u, double* f) {
  const double* q = x + 3;
  const double* v = x + 7;
  const double* r = x + 10;
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  f[3] = 0.5 * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = 0.5 * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = 0.5 * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = 0.5 * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  double R[9];
  rotmat(q, R);
  double ft[4], sum = 0;
  for (int j = 0; j < 4; ++j) { ft[j] = u[j] * 1.0 * m.tmax; }
  sum = ft[0] + ft[1];  // np.sum pairwise for 4 elems == sequential
  sum += ft[2];
  sum += ft[3];
  const double ab[3] = {0.0 / m.mass, 0.0 / m.mass, sum / m.mass};
  double at[3];
  rot(R, ab, at);
  // get_aero_drag, src/quad.py:256-277
  double vb[3], adb[3], adw[3];
  rot_conj(q, v, vb);
  for (int i = 0; i < 3; ++i) {
    const double sg = (vb[i] > 0) - (vb[i] < 0);
    adb[i] = -m.aero_drag * (vb[i] * vb[i]) * sg / m.mass;
    adb[i] -= m.rotor_drag[i] * vb[i] / m.mass;
  }
  rot(R, adb, adw);
  const double gv[3] = {0, 0, m.g};
  for (int i = 0; i < 3; ++i) f[7 + i] = -gv[i] + (-0.0 * gv[i] / m.mass) + adw[i] + at[i] + 0.0;
  double dy = 0, dx = 0, dzz = 0;
  for (int j = 0; j < 4; ++j) { dy += ft[j] * m.yf[j]; dx += ft[j] * m.xf[j]; dzz += ft[j] * m.zl[j]; }
  f[10] = 1 / m.J[0] * (dy + 0.0 + (m.J[1] - m.J[2]) * r[1] * r[2]);
  f[11] = 1 / m.J[1] * (-dx + 0.0 + (m.J[2] - m.J[0]) * r[2] * r[0]);
  f[12] = 1 / m.J[2] * (dzz + 0.0 + (m.J[0] - m.J[1]) * r[0] * r[1]);
}
// Quadrotor3D.update -> one_step_forward, src/quad.py:166-190,234-254 (u clipped to [0,1])
void plant_update(const Model& m, double* x, const double* uin, double dt) {
  double u[4], k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
  for (int j = 0; j < 4; ++j) u[j] = std::min(1.0, std::max(0.0, uin[j]));
  plant_f(m, x, u, k1);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k1[i];
  plant_f(m, xt, u, k2);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt / 2 * k2[i];
  plant_f(m, xt, u, k3);
  for (int i = 0; i < NX; ++i) xt[i] = x[i] + dt * k3[i];
  plant_f(m, xt, u, k4);
  for (int i = 0; i < NX; ++i) x[i] = x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}
// the float `while control_time < optimization_dt` loop, src/execute_trajectory.py:232-243
int plant_control_period(const Model& m, double* x, const double* u, double control_dt, double sim_dt) {
  double t = 0;
  int n = 0;
  while (t < control_dt) { plant_update(m, x, u, sim_dt); t += sim_dt; ++n; }
  return n;
}


// ---------------------------------------------------------------- RGP.learn (src/gp/RGP.py:332-505)
// Hyper-parameter learning of the recursive GP: joint state z = [g (n), eta = (L, sigma_f, sigma_n)], one scalar
// observation (Xt, yt) per call; unscented transform over eta (7 sigma points, w0 = 0.5), Kalman update of the observable
// part [sigma_n, g_t], smoother-type update of the rest, then K_x and K_x^-1 are rebuilt from the new hyper-parameters.
// Restated operation by operation, including what the reference does NOT do: the gain Jt is evaluated once at the current
// hyper-parameters (not per sigma point), the cross-covariance C_g_eta is never updated (stays zero, so St = 0), and the
// running mean is used inside the covariance accumulation loop.  The loop of the node never calls it (offline use).
struct Learner {
  int n = 0;
  std::vector<double> X, mu_g, C_g, Kxinv;
  double mu_eta[3], C_eta[9];
};
// principal square root of a symmetric positive definite 3x3 matrix (scipy.linalg.sqrtm of n/(1-w0) C): Jacobi rotations
void sqrtm3(const double* A, double* S) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::copy(A, A + 9, a);
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p * 3 + q] == 0.0) continue;
        const double th = (a[q * 3 + q] - a[p * 3 + p]) / (2 * a[p * 3 + q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1));
        const double c = 1 / std::sqrt(t * t + 1), sn = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = a[k * 3 + p], akq = a[k * 3 + q]; a[k * 3 + p] = c * akp - sn * akq; a[k * 3 + q] = sn * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = a[p * 3 + k], aqk = a[q * 3 + k]; a[p * 3 + k] = c * apk - sn * aqk; a[q * 3 + k] = sn * apk + c * aqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = v[k * 3 + p], vkq = v[k * 3 + q]; v[k * 3 + p] = c * vkp - sn * vkq; v[k * 3 + q] = sn * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double t = 0;
      for (int k = 0; k < 3; ++k) t += v[i * 3 + k] * std::sqrt(a[k * 3 + k]) * v[j * 3 + k];
      S[i * 3 + j] = t;
    }
}
void learner_rebuild(Learner& g) {   // K_x = K(X,X) + sigma_n^2 I ; K_x^-1  (:499-500)
  const int n = g.n;
  std::vector<double> K((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) K[i * n + j] = rbf(g.X[i], g.X[j], g.mu_eta[0], g.mu_eta[1]) + (i == j ? g.mu_eta[2] * g.mu_eta[2] : 0.0);
  invert(K, n, g.Kxinv);
}
void learner_init(Learner& g, int n, const double* X, const double* theta) {   // RGP.__init__ (:126-157)
  g.n = n;
  g.X.assign(X, X + n);
  g.mu_g.assign(n, 0.0);
  for (int k = 0; k < 3; ++k) g.mu_eta[k] = theta[k];
  for (int k = 0; k < 9; ++k) g.C_eta[k] = (k % 4 == 0) ? 1.0 : 0.0;
  g.C_g.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) g.C_g[i * n + j] = rbf(X[i], X[j], theta[0], theta[1]) + (i == j ? theta[2] * theta[2] : 0.0);
  learner_rebuild(g);
}
void learner_step(Learner& g, double xt, double yt) {
  const int n = g.n, np_ = n + 4, nu = n + 2;
  const double L = g.mu_eta[0], sf = g.mu_eta[1];
  std::vector<double> ks(n), Jt(n), CJ(n);
  for (int j = 0; j < n; ++j) ks[j] = rbf(xt, g.X[j], L, sf);
  for (int j = 0; j < n; ++j) { double t = 0; for (int i = 0; i < n; ++i) t += ks[i] * g.Kxinv[i * n + j]; Jt[j] = t; }
  double Jk = 0;
  for (int j = 0; j < n; ++j) Jk += Jt[j] * ks[j];
  const double Bv = rbf(xt, xt, L, sf) - Jk;
  // sigma points of eta
  double S6[9], C6[9], Sq[9], w[7], eh[7][3];
  for (int k = 0; k < 9; ++k) C6[k] = 3.0 / (1 - 0.5) * g.C_eta[k];
  sqrtm3(C6, Sq);
  (void)S6;
  w[0] = 0.5;
  for (int k = 0; k < 3; ++k) eh[0][k] = g.mu_eta[k];
  for (int i = 0; i < 3; ++i) {
    for (int k = 0; k < 3; ++k) { eh[i + 1][k] = g.mu_eta[k] + Sq[k * 3 + i]; eh[i + 4][k] = g.mu_eta[k] - Sq[k * 3 + i]; }
    w[i + 1] = w[i + 4] = (1 - w[0]) / 6.0;
  }
  // C_p_i (the same for every sigma point): [[Cg, 0, Cg Jt'], [0, 0, 0], [Jt Cg, 0, Jt Cg Jt' + B]]
  std::vector<double> Cpi((size_t)np_ * np_, 0.0), mu_p(np_, 0.0), C_p((size_t)np_ * np_, 0.0), mpi(np_), JC(n);
  double JCJ = 0, Jmu = 0;
  for (int i = 0; i < n; ++i) { double t = 0; for (int j = 0; j < n; ++j) t += g.C_g[i * n + j] * Jt[j]; CJ[i] = t; }
  for (int j = 0; j < n; ++j) { double t = 0; for (int i = 0; i < n; ++i) t += Jt[i] * g.C_g[i * n + j]; JC[j] = t; }
  for (int j = 0; j < n; ++j) { JCJ += JC[j] * Jt[j]; Jmu += Jt[j] * g.mu_g[j]; }
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) Cpi[i * np_ + j] = g.C_g[i * n + j];
    Cpi[i * np_ + n + 3] = CJ[i];
    Cpi[(n + 3) * np_ + i] = JC[i];
  }
  Cpi[(n + 3) * np_ + n + 3] = JCJ + Bv;
  for (int i = 0; i < 7; ++i) {
    for (int k = 0; k < n; ++k) mpi[k] = g.mu_g[k];
    for (int k = 0; k < 3; ++k) mpi[n + k] = eh[i][k];
    mpi[n + 3] = Jmu;
    for (int k = 0; k < np_; ++k) mu_p[k] += w[i] * mpi[k];
    for (int a = 0; a < np_; ++a)
      for (int b = 0; b < np_; ++b) C_p[a * np_ + b] += w[i] * ((mpi[a] - mu_p[a]) * (mpi[b] - mu_p[b]) + Cpi[a * np_ + b]);
  }
  // observable o = [sigma_n, g_t] = indices nu, nu+1 ; unobservable u = [g, L, sigma_f] = indices < nu
  const double mo0 = mu_p[nu], mo1 = mu_p[nu + 1];
  const double Co00 = C_p[nu * np_ + nu], Co01 = C_p[nu * np_ + nu + 1], Co10 = C_p[(nu + 1) * np_ + nu], Co11 = C_p[(nu + 1) * np_ + nu + 1];
  const double Cy = Co11 + Co00 + mo0 * mo0;
  const double G0 = Co01 / Cy, G1 = Co11 / Cy;
  const double me0 = mo0 + G0 * (yt - mo1), me1 = mo1 + G1 * (yt - mo1);
  const double Ce00 = Co00 - G0 * Cy * G0, Ce01 = Co01 - G0 * Cy * G1, Ce10 = Co10 - G1 * Cy * G0, Ce11 = Co11 - G1 * Cy * G1;
  const double det = Co00 * Co11 - Co01 * Co10;
  const double Ci00 = Co11 / det, Ci01 = -Co01 / det, Ci10 = -Co10 / det, Ci11 = Co00 / det;
  std::vector<double> Lt((size_t)nu * 2);
  for (int a = 0; a < nu; ++a) {   // Lt = C_ou' inv(C_o): C_ou[k][a] = C_p[(nu+k)][a]
    const double c0 = C_p[nu * np_ + a], c1 = C_p[(nu + 1) * np_ + a];
    Lt[a * 2] = c0 * Ci00 + c1 * Ci10;
    Lt[a * 2 + 1] = c0 * Ci01 + c1 * Ci11;
  }
  const double D00 = Ce00 - Co00, D01 = Ce01 - Co01, D10 = Ce10 - Co10, D11 = Ce11 - Co11;
  std::vector<double> mu_z(n + 3), C_z((size_t)(n + 3) * (n + 3));
  for (int a = 0; a < nu; ++a) mu_z[a] = mu_p[a] + Lt[a * 2] * (me0 - mo0) + Lt[a * 2 + 1] * (me1 - mo1);
  mu_z[nu] = me0;
  const int nz = n + 3;
  for (int a = 0; a < nu; ++a)
    for (int b = 0; b < nu; ++b) {
      const double t0 = Lt[a * 2] * D00 + Lt[a * 2 + 1] * D10, t1 = Lt[a * 2] * D01 + Lt[a * 2 + 1] * D11;
      C_z[a * nz + b] = C_p[a * np_ + b] + t0 * Lt[b * 2] + t1 * Lt[b * 2 + 1];
    }
  for (int a = 0; a < nu; ++a) {
    C_z[a * nz + nu] = Lt[a * 2] * Ce00 + Lt[a * 2 + 1] * Ce10;   // Lt C_e h
    C_z[nu * nz + a] = Ce00 * Lt[a * 2] + Ce01 * Lt[a * 2 + 1];   // h' C_e Lt'
  }
  C_z[nu * nz + nu] = Ce00;
  for (int i = 0; i < n; ++i) {
    g.mu_g[i] = mu_z[i];
    for (int j = 0; j < n; ++j) g.C_g[i * n + j] = C_z[i * nz + j];
  }
  for (int a = 0; a < 3; ++a) {
    g.mu_eta[a] = mu_z[n + a];
    for (int b = 0; b < 3; ++b) g.C_eta[a * 3 + b] = C_z[(n + a) * nz + n + b];
  }
  learner_rebuild(g);
}
struct LearnBatch { int B = 0, n = 0; std::vector<Learner> g; };   // [B][3]

// ---------------------------------------------------------------- batched engine (state + fused step)
struct Engine {
  Model m;
  int B = 0;
  std::vector<double> X, U, mu, C, xpred_prev, cost, kkt;
  std::vector<int> has_prev, idx, status, qp_iter, finished;
  std::vector<double> traj;  // [B][Tmax][13]
  std::vector<int> tlen;
  int Tmax = 0;
  std::vector<double> yref, yrefN;  // [B][N][17], [B][13]
  std::vector<double> stats;        // [B][4]: sum e_pos^2, sum e_vel^2, steps, max e_pos^2
  bool static_gp = false;           // use_gp = 1: the GP of the model is fixed (src/quad_opt.py:228-236), no regress in the loop
};

void engine_reset(Engine& e) {
  const Model& m = e.m;
  std::fill(e.X.begin(), e.X.end(), 0.0);  // acados default iterate: zeros
  std::fill(e.U.begin(), e.U.end(), 0.0);
  std::fill(e.mu.begin(), e.mu.end(), 0.0);
  for (int b = 0; b < e.B; ++b)
    for (int d = 0; d < 3; ++d)
      std::copy(m.Kx[d].begin(), m.Kx[d].end(), e.C.begin() + ((size_t)b * 3 + d) * m.nb * m.nb);
  std::fill(e.xpred_prev.begin(), e.xpred_prev.end(), 0.0);
  std::fill(e.has_prev.begin(), e.has_prev.end(), 0);
  std::fill(e.idx.begin(), e.idx.end(), 0);
  std::fill(e.status.begin(), e.status.end(), 0);
  std::fill(e.qp_iter.begin(), e.qp_iter.end(), 0);
  std::fill(e.finished.begin(), e.finished.end(), 0);
  std::fill(e.cost.begin(), e.cost.end(), 0.0);
  std::fill(e.stats.begin(), e.stats.end(), 0.0);
}

// set_reference_trajectory, src/quad_opt.py:295-317: yref_j=[x_ref_j, u_ref], yref_N = x_ref[N-1]
void make_yref(const Model& m, const double* chunk, double* yref, double* yrefN) {
  for (int j = 0; j < m.N; ++j) {
    std::memcpy(yref + j * NY, chunk + j * NX, NX * sizeof(double));
    for (int k = 0; k < NU; ++k) yref[j * NY + NX + k] = m.uref[k];
  }
  std::memcpy(yrefN, chunk + (m.N - 1) * NX, NX * sizeof(double));
}

}  // namespace

// ================================================================ C interface (ctypes)
extern "C" {

struct orc_config {
  int32_t batch, N, nb, skip;
  double T, dt_pred;
  double mass, J[3], max_thrust, x_f[4], y_f[4], z_l_tau[4], g;
  double rotor_drag[3], aero_drag;
  double W[17], W_e[13], u_lb[4], u_ub[4], u_ref[4];
  double qp_tol;
  const double* basis;  // [3*nb]
  const double* theta;  // [3*3]: per axis L, sigma_f, sigma_n
};

void* orc_create(const orc_config* c) {
  Engine* e = new Engine;
  Model& m = e->m;
  m.N = c->N; m.T = c->T; m.nb = c->nb; m.dt_pred = c->dt_pred; m.skip = c->skip;
  m.mass = c->mass; m.tmax = c->max_thrust; m.g = c->g; m.aero_drag = c->aero_drag;
  for (int i = 0; i < 3; ++i) { m.J[i] = c->J[i]; m.rotor_drag[i] = c->rotor_drag[i]; }
  for (int i = 0; i < 4; ++i) { m.xf[i] = c->x_f[i]; m.yf[i] = c->y_f[i]; m.zl[i] = c->z_l_tau[i];
    m.ulb[i] = c->u_lb[i]; m.uub[i] = c->u_ub[i]; m.uref[i] = c->u_ref[i]; }
  for (int i = 0; i < NY; ++i) m.W[i] = c->W[i];
  for (int i = 0; i < NX; ++i) m.We[i] = c->W_e[i];
  m.qp_tol = c->qp_tol > 0 ? c->qp_tol : 1e-12;
  for (int d = 0; d < 3; ++d) {
    m.Xb[d].assign(m.nb, 0.0);
    for (int j = 0; j < m.nb; ++j) m.Xb[d][j] = c->basis[d * m.nb + j];
    if (m.nb) { m.L[d] = c->theta[3 * d]; m.sf[d] = c->theta[3 * d + 1]; m.sn[d] = c->theta[3 * d + 2]; }
  }
  if (m.nb) rgp_setup(m);
  const int B = e->B = c->batch;
  e->X.assign((size_t)B * (m.N + 1) * NX, 0.0);
  e->U.assign((size_t)B * m.N * NU, 0.0);
  e->mu.assign((size_t)B * 3 * m.nb, 0.0);
  e->C.assign((size_t)B * 3 * m.nb * m.nb, 0.0);
  e->xpred_prev.assign((size_t)B * NX, 0.0);
  e->has_prev.assign(B, 0); e->idx.assign(B, 0); e->status.assign(B, 0); e->qp_iter.assign(B, 0); e->finished.assign(B, 0);
  e->cost.assign(B, 0.0); e->kkt.assign(B, 0.0);
  e->yref.assign((size_t)B * m.N * NY, 0.0); e->yrefN.assign((size_t)B * NX, 0.0);
  e->stats.assign((size_t)B * 4, 0.0);
  e->tlen.assign(B, 0);
  engine_reset(*e);
  return e;
}
void orc_destroy(void* h) { delete (Engine*)h; }
void orc_reset(void* h) { engine_reset(*(Engine*)h); }

void orc_set_trajectories(void* h, const double* traj, const int32_t* len, int Tmax) {
  Engine& e = *(Engine*)h;
  e.Tmax = Tmax;
  e.traj.assign(traj, traj + (size_t)e.B * Tmax * NX);
  for (int b = 0; b < e.B; ++b) { e.tlen[b] = len[b]; e.idx[b] = 0; e.finished[b] = 0; }
}
void orc_set_reference(void* h, const double* yref, const double* yrefN) {
  Engine& e = *(Engine*)h;
  std::copy(yref, yref + e.yref.size(), e.yref.begin());
  std::copy(yrefN, yrefN + e.yrefN.size(), e.yrefN.begin());
}
void orc_set_params(void* h, const double* mu) { Engine& e = *(Engine*)h; std::copy(mu, mu + e.mu.size(), e.mu.begin()); }

// run_optimization for the whole batch using the stored yref / params
void orc_solve(void* h, const double* x0) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b) {
    RtiOut o = rti_solve(m, &e.X[(size_t)b * (m.N + 1) * NX], &e.U[(size_t)b * m.N * NU], x0 + (size_t)b * NX,
                         &e.yref[(size_t)b * m.N * NY], &e.yrefN[(size_t)b * NX], m.nb ? &e.mu[(size_t)b * 3 * m.nb] : nullptr);
    e.cost[b] = o.cost; e.status[b] = o.status; e.qp_iter[b] = o.qp_iter; e.kkt[b] = o.kkt;
  }
}
void orc_get_x(void* h, int stage, double* out) {
  Engine& e = *(Engine*)h;
  for (int b = 0; b < e.B; ++b) std::memcpy(out + (size_t)b * NX, &e.X[((size_t)b * (e.m.N + 1) + stage) * NX], NX * sizeof(double));
}
void orc_get_u(void* h, int stage, double* out) {
  Engine& e = *(Engine*)h;
  for (int b = 0; b < e.B; ++b) std::memcpy(out + (size_t)b * NU, &e.U[((size_t)b * e.m.N + stage) * NU], NU * sizeof(double));
}
void orc_get_cost(void* h, double* out) { Engine& e = *(Engine*)h; std::copy(e.cost.begin(), e.cost.end(), out); }
void orc_get_status(void* h, int32_t* out) { Engine& e = *(Engine*)h; std::copy(e.status.begin(), e.status.end(), out); }
void orc_get_qp_iter(void* h, int32_t* out) { Engine& e = *(Engine*)h; std::copy(e.qp_iter.begin(), e.qp_iter.end(), out); }
void orc_get_kkt(void* h, double* out) { Engine& e = *(Engine*)h; std::copy(e.kkt.begin(), e.kkt.end(), out); }

void orc_predict_nominal(void* h, const double* x, const double* u, double dt, double* out) {
  Engine& e = *(Engine*)h;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b) rk4(e.m, x + (size_t)b * NX, u + (size_t)b * NU, nullptr, dt, out + (size_t)b * NX);
}
// single-instance pieces for unit tests of kernel phases
void orc_model_f(void* h, const double* x, const double* u, const double* mu, double* f, double* Jac) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
  std::vector<double> alpha;
  const double* al = nullptr;
  if (m.nb && mu) {
    alpha.assign(3 * m.nb, 0.0);
    for (int d = 0; d < 3; ++d)
      for (int i = 0; i < m.nb; ++i)
        for (int j = 0; j < m.nb; ++j) alpha[d * m.nb + i] += m.Kxinv[d][i * m.nb + j] * mu[d * m.nb + j];
    al = alpha.data();
  }
  model_f(m, x, u, al, f, Jac);
}
void orc_rk4_sens(void* h, const double* x, const double* u, const double* mu, double hstep, double* phi, double* AB) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
  std::vector<double> alpha;
  const double* al = nullptr;
  if (m.nb && mu) {
    alpha.assign(3 * m.nb, 0.0);
    for (int d = 0; d < 3; ++d)
      for (int i = 0; i < m.nb; ++i)
        for (int j = 0; j < m.nb; ++j) alpha[d * m.nb + i] += m.Kxinv[d][i * m.nb + j] * mu[d * m.nb + j];
    al = alpha.data();
  }
  rk4_sens(m, x, u, al, hstep, phi, AB);
}
void orc_compute_a_drag(const double* x, const double* xpm1, double dt, double* vb, double* ad) { compute_a_drag(x, xpm1, dt, vb, ad); }
void orc_reference_chunk(const double* traj, int len, int idx, int N, int skip, double* out) { reference_chunk(traj, len, idx, N, skip, out); }

void orc_rgp_regress(void* h, const double* vb, const double* ad) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
  const int n = m.nb;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b)
    for (int d = 0; d < 3; ++d)
      rgp_regress_axis(m, d, vb[b * 3 + d], ad[b * 3 + d], &e.mu[((size_t)b * 3 + d) * n], &e.C[((size_t)b * 3 + d) * n * n]);
}
void orc_get_rgp(void* h, double* mu, double* C) {
  Engine& e = *(Engine*)h;
  if (mu) std::copy(e.mu.begin(), e.mu.end(), mu);
  if (C) std::copy(e.C.begin(), e.C.end(), C);
}
void orc_get_kx(void* h, double* Kx, double* Kxinv) {
  Engine& e = *(Engine*)h;
  const int n = e.m.nb;
  for (int d = 0; d < 3; ++d) {
    if (Kx) std::copy(e.m.Kx[d].begin(), e.m.Kx[d].end(), Kx + (size_t)d * n * n);
    if (Kxinv) std::copy(e.m.Kxinv[d].begin(), e.m.Kxinv[d].end(), Kxinv + (size_t)d * n * n);
  }
}

// state dump / restore for teacher-forced tests
void orc_get_state(void* h, double* X, double* U, double* mu, double* C, double* xpp, int32_t* has_prev, int32_t* idx) {
  Engine& e = *(Engine*)h;
  if (X) std::copy(e.X.begin(), e.X.end(), X);
  if (U) std::copy(e.U.begin(), e.U.end(), U);
  if (mu) std::copy(e.mu.begin(), e.mu.end(), mu);
  if (C) std::copy(e.C.begin(), e.C.end(), C);
  if (xpp) std::copy(e.xpred_prev.begin(), e.xpred_prev.end(), xpp);
  if (has_prev) std::copy(e.has_prev.begin(), e.has_prev.end(), has_prev);
  if (idx) std::copy(e.idx.begin(), e.idx.end(), idx);
}
void orc_set_state(void* h, const double* X, const double* U, const double* mu, const double* C, const double* xpp,
                   const int32_t* has_prev, const int32_t* idx) {
  Engine& e = *(Engine*)h;
  if (X) std::copy(X, X + e.X.size(), e.X.begin());
  if (U) std::copy(U, U + e.U.size(), e.U.begin());
  if (mu) std::copy(mu, mu + e.mu.size(), e.mu.begin());
  if (C) std::copy(C, C + e.C.size(), e.C.begin());
  if (xpp) std::copy(xpp, xpp + e.xpred_prev.size(), e.xpred_prev.begin());
  if (has_prev) std::copy(has_prev, has_prev + e.B, e.has_prev.begin());
  if (idx) std::copy(idx, idx + e.B, e.idx.begin());
}

// The fused control step (a1..a9 of SURVEY §8): src/mpc_controller_node.py:278-318 /
// src/execute_trajectory.py:202-258.  x_meas [B,13] -> w_out [B,4]; x_pred_out optional [B,13].
void orc_step(void* h, const double* x_meas, double* w_out, double* x_pred_out) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
  const int N = m.N, n = m.nb;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b) {
    const double* x = x_meas + (size_t)b * NX;
    std::vector<double>& chunk = work().chunk;
    chunk.assign((size_t)N * NX, 0.0);
    reference_chunk(&e.traj[(size_t)b * e.Tmax * NX], e.tlen[b], e.idx[b], N, m.skip, chunk.data());
    double* yref = &e.yref[(size_t)b * N * NY];
    double* yrefN = &e.yrefN[(size_t)b * NX];
    make_yref(m, chunk.data(), yref, yrefN);
    double* U = &e.U[(size_t)b * N * NU];
    RtiOut o = rti_solve(m, &e.X[(size_t)b * (N + 1) * NX], U, x, yref, yrefN, n ? &e.mu[(size_t)b * 3 * n] : nullptr);
    e.cost[b] = o.cost; e.status[b] = o.status; e.qp_iter[b] = o.qp_iter; e.kkt[b] = o.kkt;
    double w[NU];
    for (int k = 0; k < NU; ++k) w[k] = w_out[(size_t)b * NU + k] = U[k];  // w = w_opt[0,:]
    double xp[NX];
    rk4(m, x, w, nullptr, m.dt_pred, xp);  // quad_nominal.discrete_dynamics
    e.idx[b] += 1;
    {  // trajectory finished, src/mpc_controller_node.py:374 (EPSILON_TRAJECTORY_FINISHED = 1 m, :118), x_ref = this step's chunk
      double e2 = 0;
      for (int k = 0; k < 3; ++k) e2 += (x[k] - chunk[k]) * (x[k] - chunk[k]);
      if (e.idx[b] + 1 == e.tlen[b] && std::sqrt(e2) < 1.0) e.finished[b] = 1;
    }
    if (n && !e.static_gp) {
      const double* xpm1 = e.has_prev[b] ? &e.xpred_prev[(size_t)b * NX] : x;
      double vb[3], ad[3];
      compute_a_drag(x, xpm1, m.dt_pred, vb, ad);
      for (int d = 0; d < 3; ++d)
        rgp_regress_axis(m, d, vb[d], ad[d], &e.mu[((size_t)b * 3 + d) * n], &e.C[((size_t)b * 3 + d) * n * n]);
    }
    std::memcpy(&e.xpred_prev[(size_t)b * NX], xp, sizeof(xp));
    e.has_prev[b] = 1;
    if (x_pred_out) std::memcpy(x_pred_out + (size_t)b * NX, xp, sizeof(xp));
    // tracking statistic, src/Visualiser.py:787-789,809-811: e = x_ref[0,:3] - x[:3]
    double ep = 0, ev = 0;
    for (int k = 0; k < 3; ++k) {
      ep += (x[k] - chunk[k]) * (x[k] - chunk[k]);
      ev += (x[7 + k] - chunk[7 + k]) * (x[7 + k] - chunk[7 + k]);
    }
    double* st = &e.stats[(size_t)b * 4];
    st[0] += ep; st[1] += ev; st[2] += 1.0; st[3] = std::max(st[3], ep);
  }
}
void orc_get_tracking_stats(void* h, double* out4) {
  Engine& e = *(Engine*)h;
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  for (int b = 0; b < e.B; ++b) {
    out4[0] += e.stats[b * 4]; out4[1] += e.stats[b * 4 + 1]; out4[2] += e.stats[b * 4 + 2];
    out4[3] = std::max(out4[3], e.stats[b * 4 + 3]);
  }
}
// plant: x [B,13] in place
void orc_plant_update(void* h, double* x, const double* u, double dt) {
  Engine& e = *(Engine*)h;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b) plant_update(e.m, x + (size_t)b * NX, u + (size_t)b * NU, dt);
}
int orc_plant_control_period(void* h, double* x, const double* u, double control_dt, double sim_dt) {
  Engine& e = *(Engine*)h;
  int n = 0;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < e.B; ++b) {
    int k = plant_control_period(e.m, x + (size_t)b * NX, u + (size_t)b * NU, control_dt, sim_dt);
    if (b == 0) n = k;
  }
  return n;
}

// trajectory-finished flags (src/mpc_controller_node.py:374) and the command mapping of publish_control_gazebo
// (src/mpc_controller_node.py:600-612) for the last solve
void orc_get_finished(void* h, int32_t* out) { Engine& e = *(Engine*)h; std::copy(e.finished.begin(), e.finished.end(), out); }
void orc_get_command(void* h, double* rotor, double* coll, double* rates) {
  Engine& e = *(Engine*)h;
  const Model& m = e.m;
  for (int b = 0; b < e.B; ++b) {
    const double* w = &e.U[(size_t)b * m.N * NU];   // w = w_opt[0, :]
    double s = 0;
    for (int k = 0; k < NU; ++k) { rotor[(size_t)b * NU + k] = w[k] * m.tmax / m.mass; s += w[k]; }
    coll[b] = s * m.tmax / m.mass;
    for (int k = 0; k < 3; ++k) rates[(size_t)b * 3 + k] = e.X[((size_t)b * (m.N + 1) + 1) * NX + 10 + k];   // x_opt[1, 10:13]
  }
}
// OpenMP team size of the batched entry points (the bench sweeps it); returns the previous maximum
int orc_set_threads(int n) {
#ifdef _OPENMP
  const int prev = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return prev;
#else
  (void)n;
  return 1;
#endif
}

// use_gp = 1 (static GP in the model, src/quad_opt.py:228-236): the fused step keeps mu = the training responses fixed
void orc_set_static_gp(void* h, int on) { ((Engine*)h)->static_gp = on != 0; }

// RGP.learn for B x 3 independent recursive GPs (src/gp/RGP.py:332-505): basis [3*n], theta [9] shared initial values
void* orc_learn_create(int B, int n, const double* basis, const double* theta) {
  LearnBatch* l = new LearnBatch();
  l->B = B; l->n = n; l->g.resize((size_t)B * 3);
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < 3; ++d) learner_init(l->g[(size_t)b * 3 + d], n, basis + d * n, theta + d * 3);
  return l;
}
void orc_learn_destroy(void* h) { delete (LearnBatch*)h; }
void orc_learn_step(void* h, const double* s, const double* y) {
  LearnBatch& l = *(LearnBatch*)h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < l.B * 3; ++i) learner_step(l.g[i], s[i], y[i]);
}
void orc_learn_get(void* h, double* mu_g, double* C_g, double* mu_eta, double* C_eta, double* Kxinv) {
  LearnBatch& l = *(LearnBatch*)h;
  const int n = l.n;
  for (int i = 0; i < l.B * 3; ++i) {
    const Learner& g = l.g[i];
    if (mu_g) std::copy(g.mu_g.begin(), g.mu_g.end(), mu_g + (size_t)i * n);
    if (C_g) std::copy(g.C_g.begin(), g.C_g.end(), C_g + (size_t)i * n * n);
    if (mu_eta) std::copy(g.mu_eta, g.mu_eta + 3, mu_eta + (size_t)i * 3);
    if (C_eta) std::copy(g.C_eta, g.C_eta + 9, C_eta + (size_t)i * 9);
    if (Kxinv) std::copy(g.Kxinv.begin(), g.Kxinv.end(), Kxinv + (size_t)i * n * n);
  }
}

}  // extern "C"
