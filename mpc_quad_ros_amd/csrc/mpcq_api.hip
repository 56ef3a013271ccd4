// mpcq_api.hip — C ABI (include/mpcq.h) over the HIP kernels in mpcq_kernels.hpp.
// Host side of the engine: device allocations, precision dispatch, launches on a private
// stream, HIP-event timing, optional RCCL reduction of the swarm statistics.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MPCQ_BUILDING_LIBRARY
#ifndef MPCQ_AUTO_GROUPS   // groups of mpcq_sim_steps for a streaming batch when tune.groups = 0 (see EngineT::init)
#define MPCQ_AUTO_GROUPS 2
#endif
#include "../../include/mpcq.h"
#include "mpcq_kernels.hpp"

namespace mpcq {   // mpcq_spec.hip, one translation unit per specialised shape
template <typename T> using StepFn = void (*)(const DevModel<T>, const DevState<T>, const int);
#if defined(MPCQ_SHAPE_LIST)   // reproducer builds (tools/repro_codegen): -D'MPCQ_SHAPE_LIST(X)=X(20,10) X(20,20)'
#define MPCQ_SPEC_SHAPES(X) MPCQ_SHAPE_LIST(X)
#elif defined(MPCQ_CHECKED) || defined(MPCQ_ONE_SHAPE)   // the checked build compiles for tens of minutes per specialised shape: only the headline shape has one there (MPCQ_ONE_SHAPE: quick A/B variants, `make variant SHAPES=20_10 EXTRA=-DMPCQ_ONE_SHAPE`)
#ifndef MPCQ_ONE_N
#define MPCQ_ONE_N 20
#define MPCQ_ONE_NB 10
#endif
#define MPCQ_SPEC_SHAPES(X) X(MPCQ_ONE_N, MPCQ_ONE_NB)
#else
#define MPCQ_SPEC_SHAPES(X) X(20, 10) X(20, 20) X(50, 50)   // BASELINE configs[1] (and [3] per rank), configs[2], configs[4]
#endif
// (one more macro level so that shapes given as macros -- MPCQ_ONE_N -- are expanded before the names are pasted)
#define MPCQ_DECL_(n, nb) StepFn<double> spec_lock_f64_##n##_##nb(int layout); StepFn<float> spec_lock_f32_##n##_##nb(int layout);
#define MPCQ_DECL(n, nb) MPCQ_DECL_(n, nb)
MPCQ_SPEC_SHAPES(MPCQ_DECL)
#undef MPCQ_DECL
// specialised free-running instances: the shapes of MPCQ_SPEC_RUN_SHAPES (Makefile: SPEC_RUN_SHAPES; reproducer builds: every shape)
#if defined(MPCQ_SPEC_RUN) && !defined(MPCQ_SPEC_RUN_SHAPES)
#define MPCQ_SPEC_RUN_SHAPES(X) MPCQ_SPEC_SHAPES(X)
#endif
#ifdef MPCQ_SPEC_RUN_SHAPES
#define MPCQ_DECLR_(n, nb) StepFn<double> spec_run_f64_##n##_##nb(int layout); StepFn<float> spec_run_f32_##n##_##nb(int layout);
#define MPCQ_DECLR(n, nb) MPCQ_DECLR_(n, nb)
MPCQ_SPEC_RUN_SHAPES(MPCQ_DECLR)
#endif
}
#define MPCQ_TRY64_(n, nb_) if (N == n && nb == nb_) return mpcq::spec_lock_f64_##n##_##nb_(layout);
#define MPCQ_TRY32_(n, nb_) if (N == n && nb == nb_) return mpcq::spec_lock_f32_##n##_##nb_(layout);
#define MPCQ_TRY64(n, nb_) MPCQ_TRY64_(n, nb_)
#define MPCQ_TRY32(n, nb_) MPCQ_TRY32_(n, nb_)
static mpcq::StepFn<double> spec_step(int N, int nb, int layout, double*) {   // lockstep instance of a specialised shape (layout: lds_layout), or nullptr
  MPCQ_SPEC_SHAPES(MPCQ_TRY64)
  return nullptr;
}
static mpcq::StepFn<float> spec_step(int N, int nb, int layout, float*) {
  MPCQ_SPEC_SHAPES(MPCQ_TRY32)
  return nullptr;
}
#ifdef MPCQ_SPEC_RUN_SHAPES
#define MPCQ_TRYR_(n, nb_) if (N == n && nb == nb_) return mpcq::spec_run_f64_##n##_##nb_(layout);
#define MPCQ_TRYR(n, nb_) MPCQ_TRYR_(n, nb_)
static mpcq::StepFn<double> spec_run(int N, int nb, int layout, double*) { MPCQ_SPEC_RUN_SHAPES(MPCQ_TRYR) return nullptr; }
#define MPCQ_TRYR32_(n, nb_) if (N == n && nb == nb_) return mpcq::spec_run_f32_##n##_##nb_(layout);
#define MPCQ_TRYR32(n, nb_) MPCQ_TRYR32_(n, nb_)
static mpcq::StepFn<float> spec_run(int N, int nb, int layout, float*) { MPCQ_SPEC_RUN_SHAPES(MPCQ_TRYR32) return nullptr; }
#endif

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(MPCQ_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)

// ---- RCCL through dlopen (only loaded when a communicator is requested)
struct Id128 { char b[128]; };  // ncclUniqueId (passed by value)
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  void* l = nullptr;
  if (const char* named = getenv("MPCQ_RCCL_LIB")) {   // another file than the system's RCCL (tests: a stand-in that reduces over shared memory)
    l = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
    if (!l) return fail(MPCQ_ERR_COMM, std::string("dlopen MPCQ_RCCL_LIB: ") + dlerror());
  }
  if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!l) l = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!l) return fail(MPCQ_ERR_COMM, std::string("dlopen librccl.so: ") + dlerror());
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(l, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(l, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(l, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(l, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(l, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) return fail(MPCQ_ERR_COMM, "librccl.so lacks nccl symbols");
  g_rccl.lib = l;
  return 0;
}
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2;

// Every entry point runs with the engine's device current and restores the caller's afterwards (several engines on
// different devices in one process, calls from other host threads).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// SPD inverse through Cholesky (K_x = K(X,X) + sn^2 I; np.linalg.inv in the reference, src/gp/RGP.py:157)
bool spd_inverse(const std::vector<double>& A, int n, std::vector<double>& Ai) {
  std::vector<double> G(A);
  for (int j = 0; j < n; ++j) {
    double d = G[j * n + j];
    for (int k = 0; k < j; ++k) d -= G[j * n + k] * G[j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    G[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = G[i * n + j];
      for (int k = 0; k < j; ++k) s -= G[i * n + k] * G[j * n + k];
      G[i * n + j] = s / d;
    }
  }
  std::vector<double> Gi(n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    Gi[j * n + j] = 1.0 / G[j * n + j];
    for (int i = j + 1; i < n; ++i) {
      double s = 0;
      for (int k = j; k < i; ++k) s -= G[i * n + k] * Gi[k * n + j];
      Gi[i * n + j] = s / G[i * n + i];
    }
  }
  Ai.assign(n * n, 0.0);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b <= a; ++b) {
      double s = 0;
      for (int k = a; k < n; ++k) s += Gi[k * n + a] * Gi[k * n + b];
      Ai[a * n + b] = Ai[b * n + a] = s;
    }
  return true;
}

}  // namespace

// ------------------------------------------------------------------ engine
struct mpcq_engine {
  virtual ~mpcq_engine() {}
  mpcq_config cfg;
  std::vector<double> basis, theta;
  int B = 0, N = 0, nb = 0, threads = 64;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_time = 0;
  bool have_traj = false, timed = false;
  bool tuning_env = false;   // MPCQ_TUNING=1: measurement scripts may override tuning fields through the environment
  void* comm = nullptr;
  bool comm_borrowed = false;   // comm belongs to another engine of this process (mpcq_comm_share)
  int nranks = 1;
  double* d_stats5 = nullptr;
  int n_groups = 1;              // groups of mpcq_sim_steps (mpcq_tuning.groups; EngineT::init)
  std::vector<hipEvent_t> kev;   // per-launch event pairs of the last sim_steps call
  double ktime = 0, kmin = 0, kmax = 0;   // HIP-event time of the timed step-kernel launches: total, fastest, slowest
  int klaunches = 0;
  virtual int init() = 0;
  virtual int reset() = 0;
  virtual int set_trajectories(const double*, const int32_t*, int32_t) = 0;
  virtual int set_reference(const double*, const double*) = 0;
  virtual int set_params(const double*) = 0;
  virtual int solve(const double*) = 0;
  virtual int get_x(int, double*) = 0;
  virtual int get_u(int, double*) = 0;
  virtual int get_cost(double*) = 0;
  virtual int get_int(int which, int32_t*) = 0;
  virtual int predict(const double*, const double*, double, double*) = 0;
  virtual int regress(const double*, const double*) = 0;
  virtual int get_rgp(double*, double*) = 0;
  virtual int step(const double*, double*, double*) = 0;
  virtual int step_device(const double*, double*) = 0;
  virtual int sim_reset(const double*) = 0;
  virtual int sim_steps(int, int, double) = 0;
  virtual int sim_run(int, int, double) = 0;
  virtual int sim_get(double*, double*) = 0;
  virtual int stats(double*) = 0;
  virtual int get_prof(unsigned long long*) = 0;
  virtual int get_order(int32_t*) = 0;
  virtual int get_command(double*, double*, double*) = 0;
  virtual int get_finished(int32_t*) = 0;
  virtual int get_chunk(double*) = 0;
  virtual int sim_plant(const double*, int, double) = 0;
  virtual int get_solver_state(int32_t*, double*, int32_t*) = 0;
  virtual int set_solver_state(const int32_t*, const double*, const int32_t*) = 0;
  virtual int get_state(double*, double*, double*, double*, double*, int32_t*, int32_t*) = 0;
  virtual int set_state(const double*, const double*, const double*, const double*, const double*, const int32_t*, const int32_t*) = 0;
};

namespace {

template <typename T>
struct EngineT : mpcq_engine {
  mpcq::DevModel<T> m;
  mpcq::DevState<T> st;
  mpcq::Lds L;
  size_t lds_bytes = 0;
  int resident_per_cu = 0;   // workgroups of the lockstep instance one CU holds at once (LDS and registers)
  void (*kstep)(const mpcq::DevModel<T>, const mpcq::DevState<T>, const int) = nullptr;
  void (*krun)(const mpcq::DevModel<T>, const mpcq::DevState<T>, const int) = nullptr;   // free-running variant (mpcq_sim_run)
  std::vector<double> hbufd;
  T *d_basis = nullptr, *d_Kxinv = nullptr, *d_Kx = nullptr;
  double* h_pin = nullptr;   // pinned staging of the host-buffer step: [x_meas B*13 | w B*4 | x_pred B*13]
  double *d_xin = nullptr, *d_uin = nullptr, *d_tmp = nullptr, *d_traj = nullptr, *d_xs = nullptr, *d_vb = nullptr, *d_ad = nullptr;
  int* d_tlen = nullptr;
  int* d_order = nullptr;    // launch order of the lockstep periods (order_kernel); used when the batch exceeds what the device holds at once
  bool use_order = false;
  bool split_plant = false;   // the plant update between two lockstep periods as its own launch (streaming batches), see sim_steps
  // mpcq_sim_steps with tune.groups > 1: the batch as n_groups contiguous groups, each advancing in lockstep on a stream of its own (sim_steps)
  std::vector<hipStream_t> gstreams;
  std::vector<hipEvent_t> gdone;
  hipEvent_t gstart = nullptr;
  double* d_cmd = nullptr;   // [B*8] rotor thrusts, collective thrust, body rates (mpcq_get_command); also the chunk read-back
  size_t cmd_elems = 0;
  std::vector<T> hbuf;
  std::vector<double> Kx;

  ~EngineT() override {
    DeviceGuard guard(cfg.device);
    void* ptrs[] = {st.qp_work, st.chk, st.finished, d_cmd, st.stage, st.X, st.U, st.mu, st.C, st.xpp, st.yref, st.yrefN, st.w, st.xpred, st.cost, st.stats, st.has_prev, st.idx,
                    st.status, st.qp_iter, d_basis, d_Kxinv, d_Kx, d_xin, d_uin, d_tmp, d_traj, d_xs, d_vb, d_ad, d_tlen, d_stats5, d_order};
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    if (h_pin) (void)hipHostFree(h_pin);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (hipEvent_t ev : kev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : gdone) (void)hipEventDestroy(ev);
    if (gstart) (void)hipEventDestroy(gstart);
    for (hipStream_t gs : gstreams) if (gs != stream) (void)hipStreamDestroy(gs);
    if (stream) (void)hipStreamDestroy(stream);
    if (comm && !comm_borrowed && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
  }

  template <typename P> int dalloc(P*& p, size_t n) {
    HIP_TRY(hipMalloc((void**)&p, (n ? n : 1) * sizeof(*p)));
    HIP_TRY(hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(*p), stream));
    return 0;
  }
  int h2q(T* dst, const double* src, size_t n) {
    hbuf.resize(n);
    for (size_t i = 0; i < n; ++i) hbuf[i] = (T)src[i];
    HIP_TRY(hipMemcpyAsync(dst, hbuf.data(), n * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int h2d(double* dst, const double* src, size_t n) {
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int d2h(double* dst, const double* src, size_t n) {
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int q2h(double* dst, const T* src, size_t n) {
    hbuf.resize(n);
    HIP_TRY(hipMemcpyAsync(hbuf.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (size_t i = 0; i < n; ++i) dst[i] = (double)hbuf[i];
    return 0;
  }

  int init() override {
    const mpcq_config& c = cfg;
    HIP_TRY(hipSetDevice(c.device));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    std::memset(&m, 0, sizeof(m));
    std::memset(&st, 0, sizeof(st));
    m.N = N; m.nb = nb; m.skip = c.skip; m.Tmax = 0; m.B = B;
    const bool f32 = sizeof(T) == 4;
    m.qp_max_iter = c.qp_max_iter > 0 ? c.qp_max_iter : 60;
    // (float: the interior point is not asked for more than 1e-5 -- below that its complementarity collapses while the float residual
    //  stalls, and a stage Hessian loses definiteness or the iterate its finiteness; the answer's accuracy comes from the refinement
    //  against fp64 residuals behind it, not from the interior point's own tolerance.  Smaller values are raised to 1e-5.)
    m.qp_tol = (T)(c.qp_tol > 0 ? c.qp_tol : (f32 ? 1e-5 : 1e-11));
    if (f32 && m.qp_tol < (T)1e-5) m.qp_tol = (T)1e-5;
    m.eps = f32 ? (T)6e-8 : (T)1.1e-16;
    // ---- solver tuning: mpcq_config.tune (0 = default), validated by mpcq_create_sized; with MPCQ_TUNING=1 the environment
    // overrides a field (measurement scripts).  How the defaults were measured: DESIGN.md section 3.3.
    const mpcq_tuning& tu = c.tune;
    const bool env = tuning_env = getenv("MPCQ_TUNING") && atoi(getenv("MPCQ_TUNING")) != 0;
    if (!env) {   // advisor finding: a measurement script that forgets MPCQ_TUNING=1 would otherwise compare identical configurations
      static const char* const knobs[] = {"MPCQ_WARM_MAX", "MPCQ_WARM_RETRY", "MPCQ_FLIP_MAX", "MPCQ_ABORT_PINS", "MPCQ_ABORT_WRONG", "MPCQ_POLISH_MAX", "MPCQ_PIN_RATIO",
                                          "MPCQ_IPM_MU0", "MPCQ_IPM_MARGIN", "MPCQ_IPM_TOL", "MPCQ_STAGE_MEM", "MPCQ_GENERIC", "MPCQ_BLOCK_ORDER", "MPCQ_KEV_STRIDE", "MPCQ_SPLIT_PLANT", "MPCQ_GROUPS"};
      static bool warned = false;
      for (const char* k : knobs)
        if (!warned && getenv(k)) { fprintf(stderr, "mpcq: %s is set but MPCQ_TUNING=1 is not: the environment is ignored (use mpcq_config.tune)\n", k); warned = true; }
    }
    auto ienv = [&](const char* name, int v) { const char* t = env ? getenv(name) : nullptr; return t ? atoi(t) : v; };
    auto fenv = [&](const char* name, double v) { const char* t = env ? getenv(name) : nullptr; return t ? atof(t) : v; };
    auto off = [](int v) { return v < 0 ? 0 : v; };   // -1 = "never": the kernel's encoding is 0 (flip_max: -1)
    // (float: 1e-5.  The mixed-precision method judges multipliers on double residuals, so a hand-over as tight as fp64's pays -- bench
    //  workload, 200 lockstep periods: 1e-4 / pin_ratio 1 2.70 M steps/s, 1e-5 / 0.2 2.89 M, 1e-6 / 0.2 2.94 M -- but below 1e-5 the float
    //  interior point itself loses a stage Hessian's definiteness now and then at N = 50 (the active-set method then takes over from its
    //  last iterate, solve_qp): 1e-5 keeps it out of that regime)
    m.ipm_tol = (T)fenv("MPCQ_IPM_TOL", tu.ipm_tol > 0 ? tu.ipm_tol : (f32 ? 1e-5 : 1e-6));
    // fp64 passes alternate between a multiplier check and an affine solve: twice the count of fp32's
    m.polish_max = ienv("MPCQ_POLISH_MAX", tu.polish_max ? off(tu.polish_max) : (f32 ? 12 : 16));
    // passes of the warm active-set attempt before falling back to the interior point (fp64: one factorisation each, an
    // interior-point solve costs about 15 of them)
    // 12 passes in both precisions since the factorisation stage of a pass costs little more than an interior-point one (round 4: 534 -> ~330
    // instructions): 6 -> 12 is + 3-6 % on the bench workload (four seeds), + 23 % on the round-1 spline flights 150 periods in; 16 / 24 add nothing
    m.warm_max = ienv("MPCQ_WARM_MAX", tu.warm_max > 0 ? tu.warm_max : 12);
    m.warm_retry = ienv("MPCQ_WARM_RETRY", tu.warm_retry > 0 ? tu.warm_retry : 1);
    m.ipm_margin = (T)fenv("MPCQ_IPM_MARGIN", tu.ipm_margin > 0 ? tu.ipm_margin : 0.1);
    // hand-over from the interior point: an input joins the working set when its multiplier exceeds pin_ratio x its slack
    // (weakly active ones are pinned at once, wrongly pinned ones leave through the multiplier check of the same pass)
    m.pin_ratio = (T)fenv("MPCQ_PIN_RATIO", tu.pin_ratio > 0 ? tu.pin_ratio : 0.2);
    // complementarity of the interior start in units of the gradient scale: the start is the previous solution pushed inside
    // the box, i.e. close to the new optimum, a small value makes it a warm start
    m.ipm_mu0 = (T)fenv("MPCQ_IPM_MU0", tu.ipm_mu0 > 0 ? tu.ipm_mu0 : 1e-4);
    // a fallback solve whose solution changed more than this many bound states against the previous one marks a quadrotor whose
    // saturated inputs flip between rotors every period: its next solve goes to the interior point directly
    m.flip_max = ienv("MPCQ_FLIP_MAX", tu.flip_max ? tu.flip_max : 2);
    // early exits of the warm attempt: a first pass that pins >= abort_pins inputs, a multiplier check with >= abort_wrong
    // wrong signs (the most aggressive pair that costs the round-1 spline flights nothing)
    m.abort_pins = ienv("MPCQ_ABORT_PINS", tu.abort_pins ? off(tu.abort_pins) : (N > 20 ? N / 2 : 10));
    m.abort_wrong = ienv("MPCQ_ABORT_WRONG", tu.abort_wrong ? off(tu.abort_wrong) : (N > 20 ? (9 * N) / 20 : 9));
    if (m.ipm_tol < m.qp_tol) m.ipm_tol = m.qp_tol;
    m.h = c.T / c.N; m.dt_pred = c.dt_pred;
    m.finish_r = c.finish_radius > 0 ? c.finish_radius : 1.0;
    m.mass = c.mass; m.tmax = c.max_thrust; m.g = c.g; m.aero_drag = c.aero_drag;
    for (int i = 0; i < 3; ++i) { m.J[i] = c.J[i]; m.iJ[i] = 1.0 / c.J[i]; m.rotor_drag[i] = c.rotor_drag[i]; }
    m.imass = 1.0 / c.mass;
    for (int i = 0; i < 4; ++i) {
      m.xf[i] = c.x_f[i]; m.yf[i] = c.y_f[i]; m.zl[i] = c.z_l_tau[i];
      m.ulb[i] = c.u_lb[i]; m.uub[i] = c.u_ub[i]; m.uref[i] = c.u_ref[i];
    }
    for (int i = 0; i < 17; ++i) m.W[i] = c.W[i];
    for (int i = 0; i < 13; ++i) m.We[i] = c.W_e[i];
    // RGP constants: K_x = K(X,X) + sn^2 I and its inverse (RGP.__init__, src/gp/RGP.py:140-157)
    Kx.assign((size_t)3 * nb * nb, 0.0);
    std::vector<double> Kxinv((size_t)3 * nb * nb, 0.0);
    for (int d = 0; d < 3 && nb; ++d) {
      const double Lh = theta[3 * d], sf = theta[3 * d + 1], sn = theta[3 * d + 2];
      if (!(Lh > 0)) return fail(MPCQ_ERR_INVALID, "theta: length scale must be > 0");
      m.L2inv[d] = (T)(1.0 / (Lh * Lh)); m.sf2[d] = (T)(sf * sf); m.sn2[d] = (T)(sn * sn);
      std::vector<double> K(nb * nb), Ki;
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
          const double dl = basis[d * nb + i] - basis[d * nb + j];
          K[i * nb + j] = sf * sf * std::exp(-0.5 * dl * dl / (Lh * Lh)) + (i == j ? sn * sn : 0.0);
        }
      if (!spd_inverse(K, nb, Ki)) return fail(MPCQ_ERR_INVALID, "K_x is not positive definite");
      std::copy(K.begin(), K.end(), Kx.begin() + (size_t)d * nb * nb);
      std::copy(Ki.begin(), Ki.end(), Kxinv.begin() + (size_t)d * nb * nb);
    }
    int rc;
    if ((rc = dalloc(d_basis, 3 * nb))) return rc;
    if ((rc = dalloc(d_Kxinv, (size_t)3 * nb * nb))) return rc;
    if ((rc = dalloc(d_Kx, (size_t)3 * nb * nb))) return rc;
    if (nb) {
      if ((rc = h2q(d_basis, basis.data(), 3 * nb))) return rc;
      if ((rc = h2q(d_Kxinv, Kxinv.data(), (size_t)3 * nb * nb))) return rc;
      if ((rc = h2q(d_Kx, Kx.data(), (size_t)3 * nb * nb))) return rc;
    }
    m.basis = d_basis; m.Kxinv = d_Kxinv;
    const size_t Bz = B;
    if ((rc = dalloc(st.X, Bz * (N + 1) * 13))) return rc;
    if ((rc = dalloc(st.U, Bz * N * 4))) return rc;
    if ((rc = dalloc(st.mu, Bz * 3 * nb))) return rc;
    if ((rc = dalloc(st.C, Bz * 3 * nb * nb))) return rc;
    if ((rc = dalloc(st.xpp, Bz * 13))) return rc;
    if ((rc = dalloc(st.yref, Bz * N * 17))) return rc;
    if ((rc = dalloc(st.yrefN, Bz * 13))) return rc;
    if ((rc = dalloc(st.w, Bz * 4))) return rc;
    if ((rc = dalloc(st.xpred, Bz * 13))) return rc;
    if ((rc = dalloc(st.cost, Bz))) return rc;
    if ((rc = dalloc(st.stats, Bz * 4))) return rc;
    if ((rc = dalloc(st.has_prev, Bz))) return rc;
    if ((rc = dalloc(st.idx, Bz))) return rc;
    if ((rc = dalloc(st.status, Bz))) return rc;
    if ((rc = dalloc(st.qp_iter, Bz))) return rc;
    if ((rc = dalloc(st.qp_work, Bz))) return rc;
    if ((rc = dalloc(st.finished, Bz))) return rc;
    if ((rc = dalloc(d_tlen, Bz))) return rc;
    if ((rc = dalloc(d_xin, Bz * 13))) return rc;
    if ((rc = dalloc(d_uin, Bz * 4))) return rc;
    if ((rc = dalloc(d_tmp, Bz * 13))) return rc;
    if ((rc = dalloc(d_xs, Bz * 13))) return rc;
    if ((rc = dalloc(d_vb, Bz * 3))) return rc;
    if ((rc = dalloc(d_ad, Bz * 3))) return rc;
    if ((rc = dalloc(d_stats5, 8))) return rc;
#if defined(MPCQ_PROFILE) || defined(MPCQ_TRACE_NAN)
    if ((rc = dalloc(st.prof, Bz * mpcq::PF_N))) return rc;
#endif
#ifdef MPCQ_CHECKED
    if ((rc = dalloc(st.chk, 16))) return rc;
#endif
#ifdef MPCQ_DUMP_AT
    if ((rc = dalloc(m.dbg, Bz * 4096))) return rc;
#endif
    st.tlen = d_tlen; st.traj = nullptr; st.x_meas = d_xin;
    // Layout of the working set (mpcq::lds_layout): 0 everything in LDS | 1 the per-stage records (AB'', c, qv) in the per-instance
    // global record (L2 / MALL) | 2 "compact": the Riccati gains there as well, instances limited to 256 registers -- a second wave
    // per SIMD.  Rule: LDS when the whole batch is resident at once that way, otherwise the layout that holds more instances per CU,
    // the compact one only for batches beyond what layout 0 / 1 hold at once.  mpcq_config.tune.stage_mem overrides.
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
    const bool generic = ienv("MPCQ_GENERIC", tu.generic_kernel) != 0;
    mpcq::Lds Ls[3];
    size_t bytes[3], occ[3];
    mpcq::StepFn<T> ks[3], kr[3];
    ks[0] = &mpcq::step_kernel<mpcq::Cfg<T, false>>; ks[1] = &mpcq::step_kernel<mpcq::Cfg<T, true>>; ks[2] = &mpcq::step_kernel<mpcq::Cfg<T, true, 0, -1, false, true>>;
    // free-running launches (mpcq_sim_run): the any-shape instance, replaced below by a shape-specialised one where mpcq_spec.hip has it
    kr[0] = &mpcq::step_kernel<mpcq::Cfg<T, false, 0, -1, true>>; kr[1] = &mpcq::step_kernel<mpcq::Cfg<T, true, 0, -1, true>>;
    kr[2] = &mpcq::step_kernel<mpcq::Cfg<T, true, 0, -1, true, true>>;
    for (int l = 0; l < 3; ++l) {
      Ls[l] = mpcq::lds_layout(N, nb, l, sizeof(T) == 4);
      bytes[l] = mpcq::lds_bytes<T>(Ls[l]);
      // lockstep launches: shape-specialised instances (compile-time N and nb), from mpcq_spec.hip
      if (!generic)
        if (auto k = spec_step(N, nb, l, (T*)nullptr)) ks[l] = k;
      occ[l] = 0;
      if (bytes[l] <= 160 * 1024) {   // resident workgroups per CU: LDS and registers of the instance that would run
        int nblk = 0;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ks[l]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes[l]));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(ks[l]), 64, bytes[l]) == hipSuccess && nblk > 0) occ[l] = (size_t)nblk;
        else occ[l] = (160 * 1024) / bytes[l];
      }
    }
    int layout = (occ[0] == 0 || (occ[1] > occ[0] && (size_t)B > occ[0] * n_cu)) ? 1 : 0;
    // (the compact instance is a few per cent slower per wave -- 256 registers, gains through L2 -- and adds memory-side traffic, so it
    //  takes a long stream of workgroups to pay.  Measured at N = 20, nb = 10, M steps/s layout 1 / compact: B = 2 048 5.03 / 4.83, 3 072
    //  7.04 / 6.68, 4 096 8.61 / 8.22, 6 144 9.91 / 10.49, 8 192 10.5 / 11.7, 16 384 9.5 / 10.6: from five rounds of layout 1 on.
    //  N = 50, nb = 50 (one against two per CU): B = 4 096 0.70 / 1.22.)
    {
      const size_t r1 = occ[layout] * n_cu;
      if (occ[2] > occ[layout] && (size_t)B >= 5 * r1) layout = 2;
    }
    if (tu.stage_mem) layout = tu.stage_mem - 1;
    if (const char* t = env ? getenv("MPCQ_STAGE_MEM") : nullptr) layout = (t[0] == 'c' || t[0] == 'C') ? 2 : ((t[0] == 'g' || t[0] == 'G') ? 1 : 0);
    if (occ[layout] == 0)
      return fail(MPCQ_ERR_INVALID, layout == 0 ? "per-instance working set exceeds 160 KiB LDS with the stage records in LDS (tune.stage_mem = 1)"
                                                : "per-instance working set exceeds 160 KiB LDS (N/nb too large for this precision)");
    m.gab = layout;
    const bool gab = layout != 0;
    if (env && getenv("MPCQ_VERBOSE"))
      fprintf(stderr, "mpcq: layout %d (%s), LDS %zu B per instance, %zu instances per CU; layouts 0/1/2: %zu/%zu/%zu B, %zu/%zu/%zu per CU\n", layout,
              layout == 0 ? "all LDS" : (layout == 1 ? "stage records in global memory" : "compact: stage records and gains in global memory"), bytes[layout], occ[layout],
              bytes[0], bytes[1], bytes[2], occ[0], occ[1], occ[2]);
    L = Ls[layout];
    lds_bytes = bytes[layout];
    resident_per_cu = (int)occ[layout];
    if ((rc = dalloc(st.stage, Bz * L.gtotal))) return rc;   // stage records (global placement) + multiplier rows + cost-to-go tiles (+ gains)
    // Launch order of a lockstep period: a batch beyond what the device holds at once is a stream of workgroups that ends with
    // its last one, so the quadrotors predicted to be expensive go first (mpcq::order_kernel in front of every step launch).
    // A batch that is resident as a whole starts all at once: no order needed.  tune.block_order: 1 = never, 2 = always.
    {
      const size_t resident = occ[layout] * (size_t)n_cu;
      const int bo = ienv("MPCQ_BLOCK_ORDER", tu.block_order);
      use_order = bo == 2 || (bo == 0 && (size_t)B > resident);
      if (use_order) {   // the identity until the first lockstep launch has computed an order (mpcq_get_block_order before that)
        if ((rc = dalloc(d_order, Bz))) return rc;
        std::vector<int> ident(Bz);
        for (size_t b = 0; b < Bz; ++b) ident[b] = (int)b;
        HIP_TRY(hipMemcpyAsync(d_order, ident.data(), Bz * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
      }
      // The same distinction decides where the plant update of the on-device closed loop runs (sim_steps): at the head of the next
      // step launch (one launch per period; its RK4 substeps run on one lane of every workgroup: 6 k of a quadrotor's 172 k cycles)
      // or as its own launch of one THREAD per quadrotor.  A resident batch waits for its slowest quadrotor, which the 2.5 us at
      // the head hardly move, and saves a launch; a streaming batch pays those cycles in throughput.
      split_plant = ienv("MPCQ_SPLIT_PLANT", (size_t)B > resident ? 1 : 0) != 0;
      // Groups of mpcq_sim_steps (tune.groups; 0 = automatic).  A streaming batch ends every launch with a tail in which the device
      // drains (the last workgroups finish one by one: 13 % of the resident-wave slots of a B = 8 192 launch stand empty,
      // profiles/r6_pmc_icache_b8192.json) and the next period's launch cannot start before it has: cut into groups whose launches go to
      // streams of their own, the tail of one group's period is filled by the other groups' launches.  Every group advances in lockstep,
      // quadrotors are independent, a call still ends with every quadrotor K periods on: results do not depend on the grouping.
      // A resident batch has no tail to fill (what its launch waits for is its slowest quadrotor): one group unless asked otherwise.
      n_groups = ienv("MPCQ_GROUPS", tu.groups > 0 ? tu.groups : ((size_t)B > resident ? MPCQ_AUTO_GROUPS : 1));
      if (n_groups < 1) n_groups = 1;
      while (n_groups > 1 && B / n_groups < 8) n_groups -= 1;    // (a group holds at least one quadrotor of every launch-order class)
      for (int g = 0; g < n_groups && n_groups > 1; ++g) {   // (group 0 runs on the engine's own stream: one hardware queue less)
        hipStream_t gs = stream; hipEvent_t ev;
        if (g > 0) HIP_TRY(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
        gstreams.push_back(gs);
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); gdone.push_back(ev);
      }
      if (n_groups > 1) HIP_TRY(hipEventCreateWithFlags(&gstart, hipEventDisableTiming));
    }
    kstep = ks[layout];
    krun = kr[layout];
#ifdef MPCQ_SPEC_RUN_SHAPES
    if (!generic)
      if (auto k = spec_run(N, nb, layout, (T*)nullptr)) krun = k;
#endif
    (void)gab;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(krun), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kstep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mpcq::regress_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    return reset();
  }

  int reset() override {
    const size_t Bz = B;
    HIP_TRY(hipMemsetAsync(st.X, 0, Bz * (N + 1) * 13 * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(st.U, 0, Bz * N * 4 * sizeof(double), stream));
    if (nb) HIP_TRY(hipMemsetAsync(st.mu, 0, Bz * 3 * nb * sizeof(T), stream));
    HIP_TRY(hipMemsetAsync(st.xpp, 0, Bz * 13 * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(st.stats, 0, Bz * 4 * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(st.has_prev, 0, Bz * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(st.idx, 0, Bz * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(st.status, 0, Bz * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(st.qp_iter, 0, Bz * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(st.finished, 0, Bz * sizeof(int), stream));
    if (nb) {  // C_0 = K_x for every instance and axis
      std::vector<T> h((size_t)B * 3 * nb * nb);
      for (size_t b = 0; b < Bz; ++b)
        for (size_t i = 0; i < (size_t)3 * nb * nb; ++i) h[b * 3 * nb * nb + i] = (T)Kx[i];
      HIP_TRY(hipMemcpyAsync(st.C, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }

  int set_trajectories(const double* traj, const int32_t* len, int32_t Tmax) override {
    if (Tmax <= 0) return fail(MPCQ_ERR_INVALID, "Tmax must be positive");
    for (int b = 0; b < B; ++b)
      if (len[b] <= 0 || len[b] > Tmax) return fail(MPCQ_ERR_INVALID, "trajectory length out of range");
    if (d_traj && m.Tmax != Tmax) { (void)hipFree(d_traj); d_traj = nullptr; }
    if (!d_traj) HIP_TRY(hipMalloc((void**)&d_traj, (size_t)B * Tmax * 13 * sizeof(double)));
    m.Tmax = Tmax;
    int rc;
    if ((rc = h2d(d_traj, traj, (size_t)B * Tmax * 13))) return rc;
    HIP_TRY(hipMemcpyAsync(d_tlen, len, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(st.idx, 0, (size_t)B * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(st.finished, 0, (size_t)B * sizeof(int), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    st.traj = d_traj;
    have_traj = true;
    return 0;
  }
  int set_reference(const double* yref, const double* yrefN) override {
    int rc;
    if ((rc = h2d(st.yref, yref, (size_t)B * N * 17))) return rc;
    return h2d(st.yrefN, yrefN, (size_t)B * 13);
  }
  int set_params(const double* mu) override { return nb ? h2q(st.mu, mu, (size_t)B * 3 * nb) : 0; }

  // Checked build: every call that launched the step / regress kernel ends here -- the first out-of-range index or partial EXEC
  // mask the device code recorded turns the call into an error that names it (region tag, index, valid range, lane, workgroup,
  // program counter relative to the kernel entry recorded by workgroup 0).
  int chk_after() {
#ifdef MPCQ_CHECKED
    int r[16];
    HIP_TRY(hipMemcpyAsync(r, st.chk, sizeof(r), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (r[0]) {
      HIP_TRY(hipMemsetAsync(st.chk, 0, sizeof(r), stream));
      const unsigned long long pc = ((unsigned long long)(unsigned)r[8] << 32) | (unsigned)r[7], entry = ((unsigned long long)(unsigned)r[10] << 32) | (unsigned)r[9];
      char buf[320];
      snprintf(buf, sizeof(buf), "checked build: %s tag %d index %d outside [%d, %d) on lane %d of workgroup %d, pc - entry = 0x%llx",
               r[1] >= mpcq::CK_EXEC ? "partial EXEC mask at cross-lane site," : "out-of-range access, region", r[1], r[2], r[3], r[4], r[5], r[6],
               (unsigned long long)(pc - entry));
      return fail(MPCQ_ERR_DEVICE, buf);
    }
#endif
    return 0;
  }
  int base_mode() const { return (cfg.flags & MPCQ_FLAG_STATIC_GP) ? mpcq::MODE_STATIC_GP : 0; }
  // one lockstep period of the quadrotors [b0, b0 + nq) on stream `strm`; ev_begin (if any) is recorded in front of the STEP kernel, behind
  // the ordering launch, so that the event pairs of sim_steps time the step kernel alone
  void launch_period(const mpcq::DevState<T>& s, int mode, hipEvent_t ev_begin = nullptr, hipStream_t strm = nullptr, int b0 = 0, int nq = -1) {
    if (!strm) strm = stream;
    if (nq < 0) nq = B;
    mpcq::DevState<T> so = s;
    so.b0 = b0;
    if (use_order) {   // (reads qp_iter of the previous period; a permutation by construction whatever qp_iter holds: order_bin is total)
      hipLaunchKernelGGL(mpcq::order_kernel, dim3(mpcq::ORD_CLASSES), dim3(mpcq::ORD_THREADS), mpcq::ORD_LDS, strm, (const int*)st.qp_iter + b0, nq, d_order + b0, b0);
      so.order = d_order + b0;
    }
    if (ev_begin) (void)hipEventRecord(ev_begin, strm);
    hipLaunchKernelGGL(kstep, dim3(nq), dim3(64), lds_bytes, strm, m, so, mode);
  }
  int launch_step(int mode) {
    HIP_TRY(hipEventRecord(ev0, stream));
    launch_period(st, mode);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, stream));
    timed = true;
    return 0;
  }
  int solve(const double* x0) override {
    int rc;
    if ((rc = h2d(d_xin, x0, (size_t)B * 13))) return rc;
    st.x_meas = d_xin;
    if ((rc = launch_step(0))) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    return chk_after();
  }
  int get_x(int stage, double* out) override {
    if (stage < 0 || stage > N) return fail(MPCQ_ERR_INVALID, "stage out of range");
    // one strided copy of the B rows of this stage (a reference-shaped `for i: get(i,'x')` loop stays O(N B))
    HIP_TRY(hipMemcpy2DAsync(out, 13 * sizeof(double), st.X + (size_t)stage * 13, (size_t)(N + 1) * 13 * sizeof(double), 13 * sizeof(double), B,
                             hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_u(int stage, double* out) override {
    if (stage < 0 || stage >= N) return fail(MPCQ_ERR_INVALID, "stage out of range");
    HIP_TRY(hipMemcpy2DAsync(out, 4 * sizeof(double), st.U + (size_t)stage * 4, (size_t)N * 4 * sizeof(double), 4 * sizeof(double), B,
                             hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_cost(double* out) override { return d2h(out, st.cost, B); }
  int get_int(int which, int32_t* out) override {
    const int* src = which == 0 ? st.status : (which == 1 ? st.qp_iter : (which == 2 ? st.idx : (which == 4 ? st.qp_work : st.has_prev)));
    HIP_TRY(hipMemcpyAsync(out, src, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int predict(const double* x, const double* u, double dt, double* out) override {
    int rc;
    if ((rc = h2d(d_xin, x, (size_t)B * 13))) return rc;
    if ((rc = h2d(d_uin, u, (size_t)B * 4))) return rc;
    hipLaunchKernelGGL(mpcq::predict_kernel<T>, dim3((B + 63) / 64), dim3(64), 0, stream, m, d_xin, d_uin, dt, d_tmp, B);
    HIP_TRY(hipGetLastError());
    return d2h(out, d_tmp, (size_t)B * 13);
  }
  int regress(const double* vb, const double* ad) override {
    if (!nb) return fail(MPCQ_ERR_STATE, "engine has no RGP (nb = 0)");
    int rc;
    if ((rc = h2d(d_vb, vb, (size_t)B * 3))) return rc;
    if ((rc = h2d(d_ad, ad, (size_t)B * 3))) return rc;
    hipLaunchKernelGGL(mpcq::regress_kernel<T>, dim3(B), dim3(64), lds_bytes, stream, m, st, d_vb, d_ad);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    return chk_after();
  }
  int get_rgp(double* mu, double* C) override {
    int rc;
    if (mu && nb && (rc = q2h(mu, st.mu, (size_t)B * 3 * nb))) return rc;
    if (C && nb && (rc = q2h(C, st.C, (size_t)B * 3 * nb * nb))) return rc;
    return 0;
  }
  int step(const double* x_meas, double* w_out, double* x_pred_out) override {
    if (!have_traj) return fail(MPCQ_ERR_STATE, "mpcq_step needs mpcq_set_trajectories first");
    // host buffers go through one pinned staging block so that the three copies are truly asynchronous and the
    // call synchronises once: H2D x_meas -> step kernel -> D2H w, x_pred
    const size_t nx = (size_t)B * 13, nw = (size_t)B * 4;
    if (!h_pin) HIP_TRY(hipHostMalloc((void**)&h_pin, (2 * nx + nw) * sizeof(double), hipHostMallocDefault));
    std::memcpy(h_pin, x_meas, nx * sizeof(double));
    HIP_TRY(hipMemcpyAsync(d_xin, h_pin, nx * sizeof(double), hipMemcpyHostToDevice, stream));
    st.x_meas = d_xin;
    int rc;
    if ((rc = launch_step(mpcq::MODE_TRAJ | mpcq::MODE_POST | base_mode()))) return rc;
    HIP_TRY(hipMemcpyAsync(h_pin + nx, st.w, nw * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (x_pred_out) HIP_TRY(hipMemcpyAsync(h_pin + nx + nw, st.xpred, nx * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(w_out, h_pin + nx, nw * sizeof(double));
    if (x_pred_out) std::memcpy(x_pred_out, h_pin + nx + nw, nx * sizeof(double));
    return chk_after();
  }
  int step_device(const double* d_x, double* d_w) override {
    if (!have_traj) return fail(MPCQ_ERR_STATE, "mpcq_step_device_async needs mpcq_set_trajectories first");
    mpcq::DevState<T> s2 = st;   // measurement and control are float64 in every precision (DevState::x_meas / w)
    s2.x_meas = d_x;
    s2.w_ext = d_w;   // the engine's own control record st.w is written as well (mpcq_get_command, mpcq_sim_plant_period(w = NULL))
    HIP_TRY(hipEventRecord(ev0, stream));
    launch_period(s2, mpcq::MODE_TRAJ | mpcq::MODE_POST | base_mode());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, stream));
    timed = true;
    return chk_after();   // (synchronises in the checked build only)
  }
  int sim_reset(const double* x0) override { return h2d(d_xs, x0, (size_t)B * 13); }
  int sim_steps(int K, int n_sub, double sim_dt) override {
    if (!have_traj) return fail(MPCQ_ERR_STATE, "mpcq_sim_steps needs mpcq_set_trajectories first");
    mpcq::DevState<T> s2 = st;
    s2.x_meas = d_xs;
    // HIP events around every 4th step-kernel launch: an event pair between two dependent launches costs dispatch overlap (measured
    // at B = 1 024, 20 launches: pairs on every launch 3.18 M steps/s, on every 2nd 3.22, every 4th 3.25, none 3.26), so only a sample of
    // the launches carries one (calls of fewer than 8 periods: every launch; MPCQ_KEV_STRIDE under MPCQ_TUNING=1 overrides)
    int stride = K < 8 ? 1 : 4;
    if (const char* t = tuning_env ? getenv("MPCQ_KEV_STRIDE") : nullptr) stride = atoi(t) > 0 ? atoi(t) : 1;
    const int nev = (K + stride - 1) / stride;
    while ((int)kev.size() < 2 * nev) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); kev.push_back(ev); }
    HIP_TRY(hipEventRecord(ev0, stream));
    // The plant update between two control periods rides at the head of the next step launch (MODE_PLANT_FIRST), where
    // it overlaps that launch's global loads; only the update after the last period needs the plant kernel.  Streaming batches
    // (split_plant, see create): every update is a launch of the plant kernel.  Same arithmetic, same results either way.
    const bool split = split_plant;
    s2.run_x = d_xs; s2.run_steps = 1; s2.run_nsub = n_sub; s2.run_dt = sim_dt;
    const int G = n_groups;
    if (G > 1) {
      // Groups (see init): group g = quadrotors [g0, g1), its K periods {order, step, plant} on gstreams[g], issued period by period round the
      // groups so that the host feeds every queue.  The group streams start behind everything issued on the engine's stream and the
      // engine's stream continues behind all of them.  Event pairs: group 0's launches (a launch of B / G quadrotors that shares the device).
      HIP_TRY(hipEventRecord(gstart, stream));
      for (int g = 1; g < G; ++g) HIP_TRY(hipStreamWaitEvent(gstreams[g], gstart, 0));
      const int per = ((B + G - 1) / G + 7) / 8 * 8;   // quadrotors per group (the last one takes what is left)
      for (int k = 0; k < K; ++k)
        for (int g = 0; g < G; ++g) {
          const int g0 = g * per, g1 = std::min(B, g0 + per);
          if (g0 >= g1) continue;
          const bool timed_launch = g == 0 && k % stride == 0;
          const int mode = mpcq::MODE_TRAJ | mpcq::MODE_POST | base_mode() | ((!split && k > 0) ? mpcq::MODE_PLANT_FIRST : 0);
          launch_period(s2, mode, timed_launch ? kev[2 * (k / stride)] : nullptr, gstreams[g], g0, g1 - g0);
          if (timed_launch) HIP_TRY(hipEventRecord(kev[2 * (k / stride) + 1], gstreams[g]));
          if (split || k == K - 1)
            hipLaunchKernelGGL(mpcq::plant_kernel<T>, dim3((g1 - g0 + 63) / 64), dim3(64), 0, gstreams[g], m, d_xs + (size_t)g0 * 13, st.w + (size_t)g0 * 4, n_sub, sim_dt, g1 - g0);
        }
      for (int g = 1; g < G; ++g) {
        HIP_TRY(hipEventRecord(gdone[g], gstreams[g]));
        HIP_TRY(hipStreamWaitEvent(stream, gdone[g], 0));
      }
    } else
    for (int k = 0; k < K; ++k) {
      const bool timed_launch = k % stride == 0;
      const int mode = mpcq::MODE_TRAJ | mpcq::MODE_POST | base_mode() | ((!split && k > 0) ? mpcq::MODE_PLANT_FIRST : 0);
      launch_period(s2, mode, timed_launch ? kev[2 * (k / stride)] : nullptr);
      if (timed_launch) HIP_TRY(hipEventRecord(kev[2 * (k / stride) + 1], stream));
      if (split || k == K - 1)
        hipLaunchKernelGGL(mpcq::plant_kernel<T>, dim3((B + 63) / 64), dim3(64), 0, stream, m, d_xs, st.w, n_sub, sim_dt, B);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, stream));
    timed = true;
    HIP_TRY(hipStreamSynchronize(stream));
    ktime = 0; kmin = 1e30; kmax = 0;
    klaunches = nev;
    for (int k = 0; k < nev; ++k) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, kev[2 * k], kev[2 * k + 1]));
      ktime += ms * 1e-3;
      kmin = std::min(kmin, (double)ms * 1e-3); kmax = std::max(kmax, (double)ms * 1e-3);
    }
    if (!nev) kmin = 0;
    return chk_after();
  }
  int sim_run(int K, int n_sub, double sim_dt) override {
    if (!have_traj) return fail(MPCQ_ERR_STATE, "mpcq_sim_run needs mpcq_set_trajectories first");
    if (K <= 0) return 0;
    mpcq::DevState<T> s2 = st;
    s2.x_meas = d_xs;
    s2.run_x = d_xs; s2.run_steps = K; s2.run_nsub = n_sub; s2.run_dt = sim_dt;
    while ((int)kev.size() < 2) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); kev.push_back(ev); }
    HIP_TRY(hipEventRecord(ev0, stream));
    HIP_TRY(hipEventRecord(kev[0], stream));
    hipLaunchKernelGGL(krun, dim3(B), dim3(64), lds_bytes, stream, m, s2, mpcq::MODE_TRAJ | mpcq::MODE_POST | mpcq::MODE_RUN | base_mode());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(kev[1], stream));
    HIP_TRY(hipEventRecord(ev1, stream));
    timed = true;
    HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, kev[0], kev[1]));
    ktime = kmin = kmax = ms * 1e-3;
    klaunches = 1;
    return chk_after();
  }
  int sim_get(double* x, double* w) override {
    int rc;
    if ((rc = d2h(x, d_xs, (size_t)B * 13))) return rc;
    if (w && (rc = d2h(w, st.w, (size_t)B * 4))) return rc;
    return 0;
  }
  int stats(double* out5) override {
    hipLaunchKernelGGL(mpcq::stats_kernel, dim3(1), dim3(256), 5 * 256 * sizeof(double), stream, st.stats, st.status, B, d_stats5);
    HIP_TRY(hipGetLastError());
    if (out5) {
      HIP_TRY(hipMemcpyAsync(out5, d_stats5, 5 * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    return 0;
  }
  int cmd_buf(size_t n) {
    if (cmd_elems >= n) return 0;
    if (d_cmd) { (void)hipFree(d_cmd); d_cmd = nullptr; cmd_elems = 0; }
    HIP_TRY(hipMalloc((void**)&d_cmd, n * sizeof(double)));
    cmd_elems = n;
    return 0;
  }
  int get_command(double* rotor, double* coll, double* rates) override {
    int rc;
    if ((rc = cmd_buf((size_t)B * 8))) return rc;
    hipLaunchKernelGGL(mpcq::command_kernel<T>, dim3((B + 63) / 64), dim3(64), 0, stream, m, st.w, st.X, d_cmd, d_cmd + (size_t)B * 4, d_cmd + (size_t)B * 5, B);
    HIP_TRY(hipGetLastError());
    if (rotor && (rc = d2h(rotor, d_cmd, (size_t)B * 4))) return rc;
    if (coll && (rc = d2h(coll, d_cmd + (size_t)B * 4, (size_t)B))) return rc;
    if (rates && (rc = d2h(rates, d_cmd + (size_t)B * 5, (size_t)B * 3))) return rc;
    return 0;
  }
  int get_finished(int32_t* out) override {
    HIP_TRY(hipMemcpyAsync(out, st.finished, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_chunk(double* out) override {
    if (!have_traj) return fail(MPCQ_ERR_STATE, "mpcq_get_reference_chunk needs mpcq_set_trajectories first");
    int rc;
    if ((rc = cmd_buf((size_t)B * N * 13))) return rc;
    hipLaunchKernelGGL(mpcq::chunk_kernel<T>, dim3(B), dim3(64), 0, stream, m, st.traj, st.tlen, st.idx, d_cmd);
    HIP_TRY(hipGetLastError());
    return d2h(out, d_cmd, (size_t)B * N * 13);
  }
  int sim_plant(const double* w, int n_sub, double sim_dt) override {
    int rc;
    if (w && (rc = h2d(st.w, w, (size_t)B * 4))) return rc;
    hipLaunchKernelGGL(mpcq::plant_kernel<T>, dim3((B + 63) / 64), dim3(64), 0, stream, m, d_xs, st.w, n_sub, sim_dt, B);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_solver_state(int32_t* qp_iter, double* stats4, int32_t* finished) override {
    int rc;
    if (qp_iter && (rc = get_int(1, qp_iter))) return rc;
    if (stats4 && (rc = d2h(stats4, st.stats, (size_t)B * 4))) return rc;
    if (finished && (rc = get_finished(finished))) return rc;
    return 0;
  }
  int set_solver_state(const int32_t* qp_iter, const double* stats4, const int32_t* finished) override {
    int rc;
    if (qp_iter)   // decimal fields of include/mpcq.h: passes < 1000, three one-digit fields above
      for (int b = 0; b < B; ++b)
        if (qp_iter[b] < 0 || qp_iter[b] >= 1000000) return fail(MPCQ_ERR_INVALID, "mpcq_set_solver_state: qp_iter outside [0, 1e6)");
    if (finished)
      for (int b = 0; b < B; ++b)
        if (finished[b] != 0 && finished[b] != 1) return fail(MPCQ_ERR_INVALID, "mpcq_set_solver_state: finished must be 0 or 1");
    if (qp_iter) HIP_TRY(hipMemcpyAsync(st.qp_iter, qp_iter, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    if (finished) HIP_TRY(hipMemcpyAsync(st.finished, finished, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (stats4 && (rc = h2d(st.stats, stats4, (size_t)B * 4))) return rc;
    return 0;
  }
  int get_order(int32_t* out) override {   // the permutation the last lockstep launch used (identity when no order is in use)
    if (!use_order) { for (int b = 0; b < B; ++b) out[b] = b; return 0; }
    HIP_TRY(hipMemcpyAsync(out, d_order, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
  int get_prof(unsigned long long* out) override {
#if defined(MPCQ_PROFILE) || defined(MPCQ_TRACE_NAN)
    HIP_TRY(hipMemcpyAsync(out, st.prof, (size_t)B * mpcq::PF_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
#else
    (void)out;
    return fail(MPCQ_ERR_STATE, "built without MPCQ_PROFILE");
#endif
  }
  int get_state(double* X, double* U, double* mu, double* C, double* xpp, int32_t* hp, int32_t* idx) override {
    int rc;
    if (X && (rc = d2h(X, st.X, (size_t)B * (N + 1) * 13))) return rc;
    if (U && (rc = d2h(U, st.U, (size_t)B * N * 4))) return rc;
    if (mu && nb && (rc = q2h(mu, st.mu, (size_t)B * 3 * nb))) return rc;
    if (C && nb && (rc = q2h(C, st.C, (size_t)B * 3 * nb * nb))) return rc;
    if (xpp && (rc = d2h(xpp, st.xpp, (size_t)B * 13))) return rc;
    if (hp && (rc = get_int(3, hp))) return rc;
    if (idx && (rc = get_int(2, idx))) return rc;
    return 0;
  }
  int set_state(const double* X, const double* U, const double* mu, const double* C, const double* xpp, const int32_t* hp,
                const int32_t* idx) override {
    int rc;
    // a negative cursor would make the step kernel read reference rows in front of its trajectory (only the checked build would
    // notice): refused here.  A cursor at or beyond the end is legal (get_reference_chunk repeats the last row, src/utils/utils.py:897-931).
    const bool unchecked = tuning_env && getenv("MPCQ_SKIP_STATE_CHECKS");   // tests of the checked build provoke a violation this way
    if (idx && !unchecked)
      for (int b = 0; b < B; ++b)
        if (idx[b] < 0) return fail(MPCQ_ERR_INVALID, "mpcq_set_state: negative trajectory cursor");
    if (hp)
      for (int b = 0; b < B; ++b)
        if (hp[b] != 0 && hp[b] != 1) return fail(MPCQ_ERR_INVALID, "mpcq_set_state: has_prev must be 0 or 1");
    if (X && (rc = h2d(st.X, X, (size_t)B * (N + 1) * 13))) return rc;
    if (U && (rc = h2d(st.U, U, (size_t)B * N * 4))) return rc;
    if (mu && nb && (rc = h2q(st.mu, mu, (size_t)B * 3 * nb))) return rc;
    if (C && nb && (rc = h2q(st.C, C, (size_t)B * 3 * nb * nb))) return rc;
    if (xpp && (rc = h2d(st.xpp, xpp, (size_t)B * 13))) return rc;
    if (hp) HIP_TRY(hipMemcpyAsync(st.has_prev, hp, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    if (idx) HIP_TRY(hipMemcpyAsync(st.idx, idx, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
  }
};

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* mpcq_last_error(void) { return g_err.c_str(); }
#ifndef MPCQ_SRC_ID   // csrc/Makefile: the first 16 hex digits of sha256(mpcq_kernels.hpp | mpcq_api.hip | mpcq_spec.hip | Makefile | cc_checked.sh) = bench.kernel_source_sha16()
#define MPCQ_SRC_ID "unknown"
#endif
#ifdef MPCQ_CHECKED
const char* mpcq_version(void) { return "mpcq 0.6 (gfx950, CHECKED diagnostic build, source " MPCQ_SRC_ID ")"; }
#else
const char* mpcq_version(void) { return "mpcq 0.6 (gfx950, source " MPCQ_SRC_ID ")"; }
#endif

// binaries built against the 0.3 header (source callers get the header's inline, which passes their own sizeof): the 0.3 layout ends
// in front of mpcq_tuning.block_order
int mpcq_create(const mpcq_config* c, mpcq_engine** out) { return mpcq_create_sized(c, offsetof(mpcq_config, tune) + offsetof(mpcq_tuning, block_order), out); }
int mpcq_create_sized(const mpcq_config* c_in, uint64_t cfg_size, mpcq_engine** out) {
  if (!c_in || !out) return fail(MPCQ_ERR_INVALID, "null argument");
  *out = nullptr;
  // fields behind the caller's struct size take their defaults (0); everything up to and including `flags` is required
  if (cfg_size < offsetof(mpcq_config, finish_radius) || cfg_size > sizeof(mpcq_config))
    return fail(MPCQ_ERR_INVALID, "mpcq_config size not understood by this library (built against a different mpcq.h?)");
  mpcq_config cc;
  std::memset(&cc, 0, sizeof(cc));
  std::memcpy(&cc, c_in, (size_t)cfg_size);
  const mpcq_config* c = &cc;
  if (c->batch <= 0 || c->N < 2 || c->N > 128 || c->nb < 0 || c->nb > 128 || c->skip < 1 || !(c->T > 0) || !(c->dt_pred > 0))
    return fail(MPCQ_ERR_INVALID, "bad batch/N/nb/skip/T/dt_pred");
  if (c->nb > 0 && (!c->basis || !c->theta)) return fail(MPCQ_ERR_INVALID, "nb > 0 needs basis and theta");
  if (!(c->mass > 0) || !(c->J[0] > 0) || !(c->J[1] > 0) || !(c->J[2] > 0)) return fail(MPCQ_ERR_INVALID, "bad mass/inertia");
  for (int i = 0; i < 4; ++i)
    if (!(c->u_ub[i] > c->u_lb[i])) return fail(MPCQ_ERR_INVALID, "u_ub must exceed u_lb");
  for (int i = 13; i < 17; ++i)
    if (!(c->W[i] > 0)) return fail(MPCQ_ERR_INVALID, "input weights must be positive (strictly convex QP)");
  if (!(c->finish_radius >= 0)) return fail(MPCQ_ERR_INVALID, "finish_radius must be >= 0 (0 = default 1 m)");
  if (!(c->qp_tol >= 0) || c->qp_tol > 1e-1) return fail(MPCQ_ERR_INVALID, "qp_tol out of range [0, 1e-1]");
  if (c->qp_max_iter < 0 || c->qp_max_iter > 500) return fail(MPCQ_ERR_INVALID, "qp_max_iter out of range [0, 500]");
  {
    const mpcq_tuning& t = c->tune;
    auto irange = [](int v, int lo, int hi, bool neg1) { return v == 0 || (neg1 && v == -1) || (v >= lo && v <= hi); };
    auto frange = [](double v, double lo, double hi) { return v == 0 || (v >= lo && v <= hi); };   // NaN fails both
    if (!irange(t.warm_max, 1, 64, false) || !irange(t.warm_retry, 1, 64, false) || !irange(t.flip_max, 1, 512, true) ||
        !irange(t.abort_pins, 1, 512, true) || !irange(t.abort_wrong, 1, 512, true) || !irange(t.polish_max, 1, 64, true) ||
        !irange(t.stage_mem, 1, 3, false) || !irange(t.generic_kernel, 1, 1, false) || !irange(t.block_order, 1, 2, false) || !irange(t.groups, 1, 16, false))
      return fail(MPCQ_ERR_INVALID, "mpcq_config.tune: integer field out of range (see mpcq.h)");
    // MPCQ_PRECISION_F32 answers from the active-set method (iterate and residuals in double); without it every fallback solve would return
    // the float interior point's own answer with MPCQ_SOLVE_LOW_ACCURACY and the 1e-4 budget would not hold
    if (c->precision == MPCQ_PRECISION_F32 && t.polish_max == -1)
      return fail(MPCQ_ERR_INVALID, "mpcq_config.tune.polish_max = -1 (no active-set passes) is not available with MPCQ_PRECISION_F32");
    if (!frange(t.pin_ratio, 1e-300, 1e3) || !frange(t.ipm_mu0, 1e-12, 1.0) || !(t.ipm_margin == 0 || (t.ipm_margin > 0 && t.ipm_margin < 0.5)) ||
        !frange(t.ipm_tol, 1e-300, 1e-1))
      return fail(MPCQ_ERR_INVALID, "mpcq_config.tune: real field out of range (see mpcq.h)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MPCQ_ERR_DEVICE, "no HIP device: libmpcq has no CPU path (the CPU restatement lives in oracle/ for tests only)");
  if (c->device < 0 || c->device >= ndev) return fail(MPCQ_ERR_INVALID, "device ordinal out of range");
  DeviceGuard guard(c->device);
  mpcq_engine* e = nullptr;
  if (c->precision == MPCQ_PRECISION_F64) e = new EngineT<double>();
  else if (c->precision == MPCQ_PRECISION_F32) e = new EngineT<float>();
  else return fail(MPCQ_ERR_INVALID, "unknown precision");
  e->cfg = *c;
  e->B = c->batch; e->N = c->N; e->nb = c->nb;
  if (c->nb) { e->basis.assign(c->basis, c->basis + 3 * c->nb); e->theta.assign(c->theta, c->theta + 9); }
  e->cfg.basis = nullptr; e->cfg.theta = nullptr;
  const int rc = e->init();
  if (rc) { delete e; return rc; }
  *out = e;
  return 0;
}
int mpcq_destroy(mpcq_engine* e) { delete e; return 0; }
// null check + the engine's device made current for the duration of the call
#define ENTER(e) if (!(e)) return fail(MPCQ_ERR_INVALID, "null engine"); DeviceGuard guard_((e)->cfg.device)
int mpcq_reset(mpcq_engine* e) { ENTER(e); return e->reset(); }
int mpcq_set_trajectories(mpcq_engine* e, const double* t, const int32_t* len, int32_t Tmax) { ENTER(e); if (!t || !len) return fail(MPCQ_ERR_INVALID, "null argument"); return e->set_trajectories(t, len, Tmax); }
int mpcq_set_reference(mpcq_engine* e, const double* y, const double* yN) { ENTER(e); if (!y || !yN) return fail(MPCQ_ERR_INVALID, "null argument"); return e->set_reference(y, yN); }
int mpcq_set_params(mpcq_engine* e, const double* mu) { ENTER(e); if (!mu && e->nb) return fail(MPCQ_ERR_INVALID, "null argument"); return e->set_params(mu); }
int mpcq_solve(mpcq_engine* e, const double* x0) { ENTER(e); if (!x0) return fail(MPCQ_ERR_INVALID, "x_init has to be set before running the optimization"); return e->solve(x0); }
int mpcq_get_x(mpcq_engine* e, int32_t s, double* o) { ENTER(e); return e->get_x(s, o); }
int mpcq_get_u(mpcq_engine* e, int32_t s, double* o) { ENTER(e); return e->get_u(s, o); }
int mpcq_get_cost(mpcq_engine* e, double* o) { ENTER(e); return e->get_cost(o); }
int mpcq_get_status(mpcq_engine* e, int32_t* o) { ENTER(e); return e->get_int(0, o); }
int mpcq_get_qp_iter(mpcq_engine* e, int32_t* o) { ENTER(e); return e->get_int(1, o); }
int mpcq_get_qp_work(mpcq_engine* e, int32_t* o) { ENTER(e); if (!o) return fail(MPCQ_ERR_INVALID, "null argument"); return e->get_int(4, o); }
int mpcq_get_stats(mpcq_engine* e, double* t) {
  ENTER(e);
  if (e->timed) {
    float ms = 0;
    if (hipEventSynchronize(e->ev1) == hipSuccess && hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) e->last_time = ms * 1e-3;
  }
  if (t) *t = e->last_time;
  return 0;
}
int mpcq_predict_nominal(mpcq_engine* e, const double* x, const double* u, double dt, double* o) { ENTER(e); return e->predict(x, u, dt, o); }
int mpcq_rgp_regress(mpcq_engine* e, const double* vb, const double* ad) { ENTER(e); return e->regress(vb, ad); }
int mpcq_get_rgp(mpcq_engine* e, double* mu, double* C) { ENTER(e); return e->get_rgp(mu, C); }
int mpcq_step(mpcq_engine* e, const double* x, double* w, double* xp) { ENTER(e); if (!x || !w) return fail(MPCQ_ERR_INVALID, "null argument"); return e->step(x, w, xp); }
int mpcq_step_device_async(mpcq_engine* e, const double* dx, double* dw) { ENTER(e); if (!dx) return fail(MPCQ_ERR_INVALID, "null argument"); return e->step_device(dx, dw); }
int mpcq_synchronize(mpcq_engine* e) { ENTER(e); HIP_TRY(hipStreamSynchronize(e->stream)); return 0; }
void* mpcq_stream(mpcq_engine* e) { return e ? (void*)e->stream : nullptr; }
int mpcq_get_command(mpcq_engine* e, double* rotor, double* coll, double* rates) { ENTER(e); return e->get_command(rotor, coll, rates); }
int mpcq_get_finished(mpcq_engine* e, int32_t* o) { ENTER(e); if (!o) return fail(MPCQ_ERR_INVALID, "null argument"); return e->get_finished(o); }
int mpcq_get_reference_chunk(mpcq_engine* e, double* o) { ENTER(e); if (!o) return fail(MPCQ_ERR_INVALID, "null argument"); return e->get_chunk(o); }
int mpcq_sim_reset(mpcq_engine* e, const double* x0) { ENTER(e); return e->sim_reset(x0); }
int mpcq_sim_steps(mpcq_engine* e, int32_t K, int32_t n_sub, double sim_dt) { ENTER(e); return e->sim_steps(K, n_sub, sim_dt); }
int mpcq_sim_run(mpcq_engine* e, int32_t K, int32_t n_sub, double sim_dt) { ENTER(e); return e->sim_run(K, n_sub, sim_dt); }
// `while control_time < optimization_dt: quad.update(w, simulation_dt); control_time += simulation_dt`
// (src/execute_trajectory.py:232-243): the count comes out of the same double accumulation (20 / 11 / 4 at 0.1 / 0.05 / 0.02)
int mpcq_plant_substeps(double control_dt, double sim_dt) {
  if (!(sim_dt > 0) || !(control_dt > 0) || control_dt / sim_dt > 1e6) return -1;
  double t = 0;
  int n = 0;
  while (t < control_dt) { t += sim_dt; ++n; }
  return n;
}
int mpcq_sim_plant_period(mpcq_engine* e, const double* w, double control_dt, double sim_dt, int32_t* n_sub) {
  ENTER(e);
  const int n = mpcq_plant_substeps(control_dt, sim_dt);
  if (n < 0) return fail(MPCQ_ERR_INVALID, "bad control_dt / sim_dt");
  if (n_sub) *n_sub = n;
  return e->sim_plant(w, n, sim_dt);
}
int mpcq_sim_control_periods(mpcq_engine* e, int32_t K, double control_dt, double sim_dt, int32_t* n_sub) {
  ENTER(e);
  const int n = mpcq_plant_substeps(control_dt, sim_dt);
  if (n < 0) return fail(MPCQ_ERR_INVALID, "bad control_dt / sim_dt");
  if (n_sub) *n_sub = n;
  return e->sim_steps(K, n, sim_dt);
}
int mpcq_sim_get_state(mpcq_engine* e, double* x, double* w) { ENTER(e); return e->sim_get(x, w); }
int mpcq_get_kernel_time(mpcq_engine* e, double* s, int32_t* n) { ENTER(e); if (s) *s = e->ktime; if (n) *n = e->klaunches; return 0; }
int mpcq_get_kernel_time_minmax(mpcq_engine* e, double* mn, double* mx) { ENTER(e); if (mn) *mn = e->kmin; if (mx) *mx = e->kmax; return 0; }
int mpcq_get_tracking_stats(mpcq_engine* e, double out[5]) { ENTER(e); return e->stats(out); }
#ifdef MPCQ_DUMP_AT   /* reproducer builds only (tools/repro_codegen; not part of include/mpcq.h): [B][4096] doubles */
int mpcq_debug_dump(mpcq_engine* e, double* out) {
  ENTER(e);
  double* src = e->cfg.precision == MPCQ_PRECISION_F64 ? static_cast<EngineT<double>*>(e)->m.dbg : static_cast<EngineT<float>*>(e)->m.dbg;
  HIP_TRY(hipMemcpy(out, src, (size_t)e->B * 4096 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
#endif
/* diagnostic build only: per-instance phase cycle totals of the last step, [B][16] */
int mpcq_debug_profile(mpcq_engine* e, unsigned long long* out) { ENTER(e); return e->get_prof(out); }
int mpcq_get_groups(mpcq_engine* e, int32_t* out) { ENTER(e); if (!out) return fail(MPCQ_ERR_INVALID, "null argument"); *out = e->n_groups; return 0; }
int mpcq_get_block_order(mpcq_engine* e, int32_t* out) { ENTER(e); if (!out) return fail(MPCQ_ERR_INVALID, "null argument"); return e->get_order(out); }

int mpcq_comm_unique_id(void* id128) {
  int rc = rccl_load();
  if (rc) return rc;
  const int r = g_rccl.GetUniqueId(id128);
  if (r) return fail(MPCQ_ERR_COMM, std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
  return 0;
}
int mpcq_comm_init(mpcq_engine* e, int32_t rank, int32_t nranks, const void* id128) {
  ENTER(e);
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(MPCQ_ERR_INVALID, "bad rank / nranks / id");
  if (e->comm) return fail(MPCQ_ERR_STATE, "communicator already initialised");
  int rc = rccl_load();
  if (rc) return rc;
  Id128 id;
  std::memcpy(&id, id128, sizeof(id));
  const int r = g_rccl.CommInitRank(&e->comm, nranks, id, rank);
  if (r) { e->comm = nullptr; return fail(MPCQ_ERR_COMM, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); }
  e->nranks = nranks;
  return 0;
}
int mpcq_comm_share(mpcq_engine* e, mpcq_engine* owner) {
  ENTER(e);
  if (!owner || owner == e) return fail(MPCQ_ERR_INVALID, "mpcq_comm_share: bad owner");
  if (e->comm) return fail(MPCQ_ERR_STATE, "communicator already initialised");
  if (!owner->comm || owner->comm_borrowed) return fail(MPCQ_ERR_STATE, "mpcq_comm_share: the owner has no communicator of its own");
  if (owner->cfg.device != e->cfg.device) return fail(MPCQ_ERR_INVALID, "mpcq_comm_share: the two engines live on different devices");
  e->comm = owner->comm;
  e->comm_borrowed = true;
  e->nranks = owner->nranks;
  return 0;
}
int mpcq_allreduce_tracking_stats(mpcq_engine* e, double out[5]) {
  ENTER(e);
  int rc = e->stats(nullptr);  // local 5-vector on the device
  if (rc) return rc;
  if (e->comm) {
    // slots 0,1,2,4 sum; slot 3 max: reduce [s0,s1,s2,0,s4] with SUM into d[0..4], [s3] with MAX into d[5]
    double* d = e->d_stats5;
    HIP_TRY(hipMemcpyAsync(d + 5, d + 3, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(d + 3, 0, sizeof(double), e->stream));
    int r = g_rccl.AllReduce(d, d, 5, NCCL_FLOAT64, NCCL_SUM, e->comm, e->stream);
    if (!r) r = g_rccl.AllReduce(d + 5, d + 5, 1, NCCL_FLOAT64, NCCL_MAX, e->comm, e->stream);
    if (r) return fail(MPCQ_ERR_COMM, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    HIP_TRY(hipMemcpyAsync(d + 3, d + 5, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  }
  HIP_TRY(hipMemcpyAsync(out, e->d_stats5, 5 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return 0;
}
int mpcq_get_state(mpcq_engine* e, double* X, double* U, double* mu, double* C, double* xpp, int32_t* hp, int32_t* idx) { ENTER(e); return e->get_state(X, U, mu, C, xpp, hp, idx); }
int mpcq_set_state(mpcq_engine* e, const double* X, const double* U, const double* mu, const double* C, const double* xpp, const int32_t* hp, const int32_t* idx) { ENTER(e); return e->set_state(X, U, mu, C, xpp, hp, idx); }
int mpcq_get_solver_state(mpcq_engine* e, int32_t* qp_iter, double* stats4, int32_t* finished) { ENTER(e); return e->get_solver_state(qp_iter, stats4, finished); }
int mpcq_set_solver_state(mpcq_engine* e, const int32_t* qp_iter, const double* stats4, const int32_t* finished) { ENTER(e); return e->set_solver_state(qp_iter, stats4, finished); }

}  // extern "C"
