// mpcq_learn.hip — batched RGP.learn (src/gp/RGP.py:332-505): hyper-parameter learning of the recursive GP for B x 3
// independent (quadrotor, axis) regressors, one 64-lane workgroup per regressor, fp64, state resident in HBM.
// SURVEY §8 f4 ("next" row): the loop body of the node never calls learn (it is the reference's offline estimator), so this
// is its own small object behind the C ABI (mpcq_learn_* in include/mpcq.h), not part of the fused control step.
//
// What one call does per regressor with the new scalar sample (s, y), in the reference's operation order:
//   Jt = k(s, X) K_x^-1, B = k(s,s) - Jt k(X,s)                                   (gain at the CURRENT hyper-parameters)
//   sigma points of eta = (L, sigma_f, sigma_n): eta_0 = mu, eta_i = mu +- sqrtm(6 C_eta)[:, i], w = (1/2, 1/12 ...)
//   p = [g, eta, g_t]: mu_p, C_p accumulated over the 7 points WITH THE RUNNING MEAN inside the loop (as the reference)
//   Kalman update of the observable part o = [sigma_n, g_t] with C_y = C_o[1,1] + C_o[0,0] + mu_o[0]^2, smoother-type
//   update of the rest u = [g, L, sigma_f] through Lt = C_ou' C_o^-1; new (mu_g, C_g, mu_eta, C_eta)
//   K_x = K(X,X) + sigma_n^2 I and K_x^-1 (Gauss-Jordan with partial pivoting) for the new hyper-parameters.
// The cross-covariance C_g_eta of the reference is never updated there (it stays zero), so the St terms vanish.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpcq.h"

namespace mpcq {
extern __shared__ unsigned char smem_raw[];

struct LearnState {
  int B, n;
  const double* X;     // [3][n] basis vectors (shared by the batch)
  double* mu_g;        // [B][3][n]
  double* C_g;         // [B][3][n][n]
  double* mu_eta;      // [B][3][3]
  double* C_eta;       // [B][3][3][3]
  double* Kxinv;       // [B][3][n][n]
};

__device__ inline double l_rbf(double x1, double x2, double L, double sf) {
  const double d = x1 - x2, invLL = 1.0 / (L * L);
  return sf * sf * exp(((-0.5 * d) * invLL) * d);
}

// principal square root of a symmetric positive definite 3x3 matrix (scipy.linalg.sqrtm in the reference): Jacobi rotations
__device__ inline void l_sqrtm3(const double* A, double* S) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) a[k] = A[k];
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p * 3 + q] == 0.0) continue;
        const double th = (a[q * 3 + q] - a[p * 3 + p]) / (2 * a[p * 3 + q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
        const double c = 1 / sqrt(t * t + 1), sn = t * c;
        for (int k = 0; k < 3; ++k) { const double x = a[k * 3 + p], y = a[k * 3 + q]; a[k * 3 + p] = c * x - sn * y; a[k * 3 + q] = sn * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = a[p * 3 + k], y = a[q * 3 + k]; a[p * 3 + k] = c * x - sn * y; a[q * 3 + k] = sn * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = v[k * 3 + p], y = v[k * 3 + q]; v[k * 3 + p] = c * x - sn * y; v[k * 3 + q] = sn * x + c * y; }
      }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double t = 0;
      for (int k = 0; k < 3; ++k) t += v[i * 3 + k] * sqrt(a[k * 3 + k]) * v[j * 3 + k];
      S[i * 3 + j] = t;
    }
}

// K_x = K(X,X) + sigma_n^2 I and its inverse into global Kxinv; M, Ai: LDS [n][n] each; one lane per row
__device__ inline void l_rebuild(const double* X, int n, double L, double sf, double sn, double* M, double* Ai, double* gout, int* piv) {
  const int t = threadIdx.x;
  for (int it = t; it < n * n; it += blockDim.x) {
    const int i = it / n, j = it - i * n;
    M[it] = l_rbf(X[i], X[j], L, sf) + (i == j ? sn * sn : 0.0);
    Ai[it] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int c = 0; c < n; ++c) {
    if (t == 0) {
      int p = c;
      for (int i = c + 1; i < n; ++i)
        if (fabs(M[i * n + c]) > fabs(M[p * n + c])) p = i;
      *piv = p;
    }
    __syncthreads();
    const int p = *piv;
    if (p != c && t < n) {
      const double a = M[p * n + t], b = M[c * n + t], ai = Ai[p * n + t], bi = Ai[c * n + t];
      M[p * n + t] = b; M[c * n + t] = a; Ai[p * n + t] = bi; Ai[c * n + t] = ai;
    }
    __syncthreads();
    const double d = 1.0 / M[c * n + c];
    __syncthreads();
    if (t < n) { M[c * n + t] *= d; Ai[c * n + t] *= d; }
    __syncthreads();
    if (t < n && t != c) {
      const double f = M[t * n + c];
      if (f != 0.0)
        for (int j = 0; j < n; ++j) { M[t * n + j] -= f * M[c * n + j]; Ai[t * n + j] -= f * Ai[c * n + j]; }
    }
    __syncthreads();
  }
  for (int it = t; it < n * n; it += blockDim.x) gout[it] = Ai[it];
}

__global__ void __launch_bounds__(64) learn_init_kernel(const LearnState st, const double* theta /*[3][3]*/) {
  const int r = blockIdx.x, d = r % 3, n = st.n, t = threadIdx.x;
  double* D = reinterpret_cast<double*>(smem_raw);
  double* M = D;
  double* Ai = D + n * n;
  int* piv = reinterpret_cast<int*>(D + 2 * n * n);
  const double* X = st.X + d * n;
  const double L = theta[d * 3], sf = theta[d * 3 + 1], sn = theta[d * 3 + 2];
  for (int it = t; it < n; it += blockDim.x) st.mu_g[(size_t)r * n + it] = 0.0;
  for (int it = t; it < n * n; it += blockDim.x) {
    const int i = it / n, j = it - i * n;
    st.C_g[(size_t)r * n * n + it] = l_rbf(X[i], X[j], L, sf) + (i == j ? sn * sn : 0.0);   // C_0 = K(X,X) + sigma_n^2 I
  }
  if (t < 3) st.mu_eta[(size_t)r * 3 + t] = theta[d * 3 + t];
  if (t < 9) st.C_eta[(size_t)r * 9 + t] = (t % 4 == 0) ? 1.0 : 0.0;
  l_rebuild(X, n, L, sf, sn, M, Ai, st.Kxinv + (size_t)r * n * n, piv);
}

__global__ void __launch_bounds__(64) learn_step_kernel(const LearnState st, const double* s_in, const double* y_in) {
  const int r = blockIdx.x, d = r % 3, n = st.n, t = threadIdx.x, NT = blockDim.x;
  const int np_ = n + 4, nu = n + 2, nz = n + 3;
  double* D = reinterpret_cast<double*>(smem_raw);
  // LDS: Cp [np][np] | W [max(7 np, 2 n n)] (running means, later the Gauss-Jordan workspace) | ks Jt CJ JC mug [n] each | Lt [nu][2] | sc [32]
  double* Cp = D;
  double* W = Cp + np_ * np_;
  const int wsz = 7 * np_ > 2 * n * n ? 7 * np_ : 2 * n * n;
  double* ks = W + wsz;
  double* Jt = ks + n;
  double* CJ = Jt + n;
  double* JC = CJ + n;
  double* mug = JC + n;
  double* Lt = mug + n;
  double* sc = Lt + 2 * nu;
  int* piv = reinterpret_cast<int*>(sc + 32);
  const double* X = st.X + d * n;
  double* gmu = st.mu_g + (size_t)r * n;
  double* gC = st.C_g + (size_t)r * n * n;
  double* geta = st.mu_eta + (size_t)r * 3;
  double* gCe = st.C_eta + (size_t)r * 9;
  const double* gKi = st.Kxinv + (size_t)r * n * n;
  const double xt = s_in[r], yt = y_in[r];
  const double e0 = geta[0], e1 = geta[1], e2 = geta[2];
  for (int j = t; j < n; j += NT) { ks[j] = l_rbf(xt, X[j], e0, e1); mug[j] = gmu[j]; }
  __syncthreads();
  for (int j = t; j < n; j += NT) { double a = 0; for (int i = 0; i < n; ++i) a += ks[i] * gKi[i * n + j]; Jt[j] = a; }
  __syncthreads();
  for (int i = t; i < n; i += NT) {
    double a = 0, b = 0;
    for (int j = 0; j < n; ++j) { a += gC[i * n + j] * Jt[j]; b += Jt[j] * gC[j * n + i]; }
    CJ[i] = a; JC[i] = b;
  }
  __syncthreads();
  if (t == 0) {
    double Jk = 0, JCJ = 0, Jmu = 0;
    for (int j = 0; j < n; ++j) { Jk += Jt[j] * ks[j]; JCJ += JC[j] * Jt[j]; Jmu += Jt[j] * mug[j]; }
    sc[0] = l_rbf(xt, xt, e0, e1) - Jk;   // B
    sc[1] = JCJ; sc[2] = Jmu;
    // sigma points: eta_hat[i][k] in sc[4 + 3 i + k]
    double C6[9], Sq[9];
    for (int k = 0; k < 9; ++k) C6[k] = 3.0 / (1 - 0.5) * gCe[k];
    l_sqrtm3(C6, Sq);
    const double mu[3] = {e0, e1, e2};
    for (int k = 0; k < 3; ++k) sc[4 + k] = mu[k];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) { sc[4 + 3 * (i + 1) + k] = mu[k] + Sq[k * 3 + i]; sc[4 + 3 * (i + 4) + k] = mu[k] - Sq[k * 3 + i]; }
  }
  __syncthreads();
  const double Bv = sc[0], JCJ = sc[1], Jmu = sc[2];
  auto w_of = [](int i) { return i == 0 ? 0.5 : (1 - 0.5) / 6.0; };
  auto mpi = [&](int i, int a) { return a < n ? mug[a] : (a < n + 3 ? sc[4 + 3 * i + (a - n)] : Jmu); };
  auto cpi = [&](int a, int b) {
    if (a < n && b < n) return gC[a * n + b];
    if (a < n && b == n + 3) return CJ[a];
    if (a == n + 3 && b < n) return JC[b];
    if (a == n + 3 && b == n + 3) return JCJ + Bv;
    return 0.0;
  };
  // running means mu_run[i][a] (the reference subtracts the mean accumulated SO FAR inside the loop)
  for (int a = t; a < np_; a += NT) {
    double m = 0;
    for (int i = 0; i < 7; ++i) { m += w_of(i) * mpi(i, a); W[i * np_ + a] = m; }
  }
  __syncthreads();
  for (int it = t; it < np_ * np_; it += NT) {
    const int a = it / np_, b = it - a * np_;
    const double c = cpi(a, b);
    double acc = 0;
    for (int i = 0; i < 7; ++i) acc += w_of(i) * ((mpi(i, a) - W[i * np_ + a]) * (mpi(i, b) - W[i * np_ + b]) + c);
    Cp[it] = acc;
  }
  __syncthreads();
  // observable o = [sigma_n, g_t] = rows nu, nu+1; every lane forms the 2x2 quantities itself
  const double mo0 = W[6 * np_ + nu], mo1 = W[6 * np_ + nu + 1];
  const double Co00 = Cp[nu * np_ + nu], Co01 = Cp[nu * np_ + nu + 1], Co10 = Cp[(nu + 1) * np_ + nu], Co11 = Cp[(nu + 1) * np_ + nu + 1];
  const double Cy = Co11 + Co00 + mo0 * mo0;
  const double G0 = Co01 / Cy, G1 = Co11 / Cy;
  const double me0 = mo0 + G0 * (yt - mo1), me1 = mo1 + G1 * (yt - mo1);
  const double Ce00 = Co00 - G0 * Cy * G0, Ce01 = Co01 - G0 * Cy * G1, Ce10 = Co10 - G1 * Cy * G0, Ce11 = Co11 - G1 * Cy * G1;
  const double det = Co00 * Co11 - Co01 * Co10;
  const double Ci00 = Co11 / det, Ci01 = -Co01 / det, Ci10 = -Co10 / det, Ci11 = Co00 / det;
  for (int a = t; a < nu; a += NT) {
    const double c0 = Cp[nu * np_ + a], c1 = Cp[(nu + 1) * np_ + a];
    Lt[a * 2] = c0 * Ci00 + c1 * Ci10;
    Lt[a * 2 + 1] = c0 * Ci01 + c1 * Ci11;
  }
  __syncthreads();
  const double D00 = Ce00 - Co00, D01 = Ce01 - Co01, D10 = Ce10 - Co10, D11 = Ce11 - Co11;
  auto mu_z = [&](int a) { return a < nu ? W[6 * np_ + a] + Lt[a * 2] * (me0 - mo0) + Lt[a * 2 + 1] * (me1 - mo1) : me0; };
  auto C_z = [&](int a, int b) {
    if (a < nu && b < nu) {
      const double t0 = Lt[a * 2] * D00 + Lt[a * 2 + 1] * D10, t1 = Lt[a * 2] * D01 + Lt[a * 2 + 1] * D11;
      return Cp[a * np_ + b] + t0 * Lt[b * 2] + t1 * Lt[b * 2 + 1];
    }
    if (a < nu) return Lt[a * 2] * Ce00 + Lt[a * 2 + 1] * Ce10;
    if (b < nu) return Ce00 * Lt[b * 2] + Ce01 * Lt[b * 2 + 1];
    return Ce00;
  };
  (void)nz;
  for (int a = t; a < n; a += NT) gmu[a] = mu_z(a);
  for (int it = t; it < n * n; it += NT) { const int a = it / n, b = it - a * n; gC[it] = C_z(a, b); }
  double ne[3];
  for (int a = 0; a < 3; ++a) ne[a] = mu_z(n + a);
  if (t < 3) geta[t] = ne[t];
  if (t < 9) gCe[t] = C_z(n + t / 3, n + t % 3);
  __syncthreads();   // W (running means) is read above; it becomes the Gauss-Jordan workspace now
  l_rebuild(X, n, ne[0], ne[1], ne[2], W, W + n * n, st.Kxinv + (size_t)r * n * n, piv);
}

}  // namespace mpcq

// ------------------------------------------------------------------ C ABI
namespace {
thread_local std::string l_err;
int lfail(int code, const std::string& msg) { l_err = msg; return code; }
#define L_TRY(expr)                                                                                  \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) return lfail(MPCQ_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
struct Guard {
  int prev = -1; bool sw = false;
  explicit Guard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) sw = hipSetDevice(dev) == hipSuccess; }
  ~Guard() { if (sw) (void)hipSetDevice(prev); }
};
}  // namespace

struct mpcq_learner {
  mpcq::LearnState st;
  int device = 0;
  size_t lds = 0;
  hipStream_t stream = nullptr;
  double *d_X = nullptr, *d_theta = nullptr, *d_s = nullptr, *d_y = nullptr;
};

extern "C" {

const char* mpcq_learn_last_error(void) { return l_err.c_str(); }

int mpcq_learn_create(int32_t batch, int32_t nb, const double* basis, const double* theta, int32_t device, mpcq_learner** out) {
  if (!out) return lfail(MPCQ_ERR_INVALID, "null argument");
  *out = nullptr;
  if (batch <= 0 || nb < 1 || nb > 64 || !basis || !theta) return lfail(MPCQ_ERR_INVALID, "bad batch / nb (1..64) / basis / theta");
  for (int d = 0; d < 3; ++d)
    if (!(theta[d * 3] > 0)) return lfail(MPCQ_ERR_INVALID, "theta: length scale must be > 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return lfail(MPCQ_ERR_DEVICE, "no HIP device: libmpcq has no CPU path");
  if (device < 0 || device >= ndev) return lfail(MPCQ_ERR_INVALID, "device ordinal out of range");
  Guard g(device);
  mpcq_learner* l = new mpcq_learner();
  l->device = device;
  const size_t R = (size_t)batch * 3, n = nb;
  std::memset(&l->st, 0, sizeof(l->st));
  l->st.B = batch; l->st.n = nb;
  auto fail_free = [&](int rc) { mpcq_learn_destroy(l); return rc; };
#define L_ALLOC(p, cnt) if (hipMalloc((void**)&(p), (cnt) * sizeof(double)) != hipSuccess) return fail_free(lfail(MPCQ_ERR_DEVICE, "hipMalloc"))
  L_ALLOC(l->d_X, 3 * n); L_ALLOC(l->d_theta, 9); L_ALLOC(l->d_s, R); L_ALLOC(l->d_y, R);
  L_ALLOC(l->st.mu_g, R * n); L_ALLOC(l->st.C_g, R * n * n); L_ALLOC(l->st.mu_eta, R * 3); L_ALLOC(l->st.C_eta, R * 9); L_ALLOC(l->st.Kxinv, R * n * n);
#undef L_ALLOC
  l->st.X = l->d_X;
  if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) return fail_free(lfail(MPCQ_ERR_DEVICE, "hipStreamCreate"));
  const size_t np_ = n + 4, wsz = std::max(7 * np_, 2 * n * n);
  l->lds = (np_ * np_ + wsz + 5 * n + 2 * (n + 2) + 32 + 2) * sizeof(double);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mpcq::learn_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l->lds) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mpcq::learn_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l->lds) != hipSuccess)
    return fail_free(lfail(MPCQ_ERR_DEVICE, "hipFuncSetAttribute"));
  if (hipMemcpyAsync(l->d_X, basis, 3 * n * sizeof(double), hipMemcpyHostToDevice, l->stream) != hipSuccess ||
      hipMemcpyAsync(l->d_theta, theta, 9 * sizeof(double), hipMemcpyHostToDevice, l->stream) != hipSuccess)
    return fail_free(lfail(MPCQ_ERR_DEVICE, "hipMemcpy"));
  hipLaunchKernelGGL(mpcq::learn_init_kernel, dim3((unsigned)R), dim3(64), l->lds, l->stream, l->st, l->d_theta);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(l->stream) != hipSuccess) return fail_free(lfail(MPCQ_ERR_DEVICE, "learn_init_kernel"));
  *out = l;
  return 0;
}

int mpcq_learn_destroy(mpcq_learner* l) {
  if (!l) return 0;
  Guard g(l->device);
  void* ptrs[] = {l->d_X, l->d_theta, l->d_s, l->d_y, l->st.mu_g, l->st.C_g, l->st.mu_eta, l->st.C_eta, l->st.Kxinv};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (l->stream) (void)hipStreamDestroy(l->stream);
  delete l;
  return 0;
}

int mpcq_learn_step(mpcq_learner* l, const double* v_body, const double* a_drag) {
  if (!l || !v_body || !a_drag) return lfail(MPCQ_ERR_INVALID, "null argument");
  Guard g(l->device);
  const size_t R = (size_t)l->st.B * 3;
  L_TRY(hipMemcpyAsync(l->d_s, v_body, R * sizeof(double), hipMemcpyHostToDevice, l->stream));
  L_TRY(hipMemcpyAsync(l->d_y, a_drag, R * sizeof(double), hipMemcpyHostToDevice, l->stream));
  hipLaunchKernelGGL(mpcq::learn_step_kernel, dim3((unsigned)R), dim3(64), l->lds, l->stream, l->st, l->d_s, l->d_y);
  L_TRY(hipGetLastError());
  L_TRY(hipStreamSynchronize(l->stream));
  return 0;
}

int mpcq_learn_get(mpcq_learner* l, double* mu_g, double* C_g, double* mu_eta, double* C_eta, double* Kx_inv) {
  if (!l) return lfail(MPCQ_ERR_INVALID, "null argument");
  Guard g(l->device);
  const size_t R = (size_t)l->st.B * 3, n = l->st.n;
  if (mu_g) L_TRY(hipMemcpyAsync(mu_g, l->st.mu_g, R * n * sizeof(double), hipMemcpyDeviceToHost, l->stream));
  if (C_g) L_TRY(hipMemcpyAsync(C_g, l->st.C_g, R * n * n * sizeof(double), hipMemcpyDeviceToHost, l->stream));
  if (mu_eta) L_TRY(hipMemcpyAsync(mu_eta, l->st.mu_eta, R * 3 * sizeof(double), hipMemcpyDeviceToHost, l->stream));
  if (C_eta) L_TRY(hipMemcpyAsync(C_eta, l->st.C_eta, R * 9 * sizeof(double), hipMemcpyDeviceToHost, l->stream));
  if (Kx_inv) L_TRY(hipMemcpyAsync(Kx_inv, l->st.Kxinv, R * n * n * sizeof(double), hipMemcpyDeviceToHost, l->stream));
  L_TRY(hipStreamSynchronize(l->stream));
  return 0;
}

}  // extern "C"
