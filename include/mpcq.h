/* mpcq.h — C ABI of libmpcq.so, the MI355X (gfx950) batched MPC+RGP control-step engine.
 *
 * Drop-in boundary.  In the reference the hot path sits behind the Python class
 * `quad_optimizer` (src/quad_opt.py:35) whose only native FFI is acados_template's ctypes binding
 * of the generated solver: `.set(stage,'yref'|'lbx'|'ubx'|'p',…)`, `.solve()`, `.get(stage,'x'|'u')`,
 * `.get_stats('time_tot')`, `.get_cost()` (src/quad_opt.py:286-290,311-315,328-333,342-350,404),
 * plus numpy code for the recursive GP (src/gp/RGP.py:303-330 through src/gp/GPE.py:244-268).
 * This header is the batched analogue: one engine = B independent quadrotors advanced in lockstep,
 * all state (SQP iterate, RGP mean/covariance, trajectory cursor) resident in HBM between calls.
 *
 * Conventions: every function returns 0 on success and a negative mpcq_status on failure
 * (mpcq_last_error() gives the message).  All host arrays are caller-owned, C-contiguous,
 * float64, batch-major [B, ...] exactly like the numpy arrays of the reference facade; the
 * engine converts to its compute precision on the device.  Calls block until the result is
 * available unless the name ends in _async.  Not thread-safe per handle (the reference drives
 * its solver from one rospy callback thread, src/mpc_controller_node.py:234).
 *
 * State layout: x = [p(3), q = (w,x,y,z)(4), v(3) world, r(3) body]  (src/quad_opt.py:168-174),
 * u in [0,1]^4 (src/quad_opt.py:142-144), y = [x, u] (17).
 */
#ifndef MPCQ_H
#define MPCQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCQ_NX 13
#define MPCQ_NU 4
#define MPCQ_NY 17

typedef enum mpcq_status {
  MPCQ_OK = 0,
  MPCQ_ERR_INVALID = -1,   /* bad argument / configuration */
  MPCQ_ERR_DEVICE = -2,    /* HIP runtime error (no GPU, OOM, launch failure) */
  MPCQ_ERR_STATE = -3,     /* call sequence error (e.g. step before set_trajectories) */
  MPCQ_ERR_COMM = -4       /* RCCL error */
} mpcq_status;

/* per-instance solver status, the acados return codes the reference discards (src/quad_opt.py:333) */
#define MPCQ_SOLVE_OK 0
#define MPCQ_SOLVE_NAN 1          /* the QP step was not finite: the instance kept its previous iterate and control */
#define MPCQ_SOLVE_MAXITER 2
#define MPCQ_SOLVE_QP_FAILURE 4
#define MPCQ_SOLVE_LOW_ACCURACY 8 /* MPCQ_PRECISION_F32 only, a warning: the step was taken, but its control may be off by more than the 1e-4 budget.
                                     Either the refinement of the QP solution against fp64 residuals did not converge (the step is the
                                     interior point's float answer), or the working set cycled under the float factorisation and the set the
                                     method settled on ignores a wrong-signed multiplier worth more than 1e-6 of a control (round 6; until then
                                     such solves came back with status 0, up to 0.43 of full thrust off on a flight that tumbles).  One in 7 M
                                     solves of the bench workload checked against the fp64 engine, on a quadrotor that is lost (QP gradient
                                     scale 1e9; DESIGN.md section 3.2, INTEGRATION.md says the same): expected where a flight is lost
                                     altogether.  Every solve that reports 0 is within the
                                     budget (tests/parity_cases.py: case_f32_every_solve_against_f64).  Not a failure:
                                     mpcq_get_tracking_stats out[4] does not count it */

/* mpcq_config.flags.  MPCQ_FLAG_STATIC_GP: the GP in the model is a static one (use_gp = 1, gpe.type == "GP",
 * src/quad_opt.py:228-236 with src/gp/GP.py:136-175): basis = its training inputs, theta = (L, sigma_f,
 * sqrt(noise + 1e-7)), the training responses are loaded once with mpcq_set_params and the fused step does NOT
 * run the recursive update (mean and covariance stay as they are). */
#define MPCQ_FLAG_STATIC_GP 1

/* MPCQ_PRECISION_F64 (default): the reference's own arithmetic; <= 1e-7 relative control deviation from the fp64 oracle.  (The interior
 * point of a fallback solve iterates in float where the shape allows, N <= 32: it only has to name a working set -- the active-set
 * method behind it, which produces the answer, and everything else are double.)
 * MPCQ_PRECISION_F32: mixed precision (round 5).  Storage and bulk arithmetic in float -- stage records (sensitivities, gaps, cost
 * gradients), Riccati factorisation on the matrix cores, gains, sweeps, RGP state --, the accuracy of double where cond(H) ~ 2e6
 * demands it: the iterate, the measurement and every difference that defines the QP are formed in double (as in F64), the shooting
 * integrates in double and rounds its RECORDS to float once, and the QP solution is kept in double and refined against the residual
 * of the QP evaluated in double on those records, the float factorisation solving for the corrections (iterative refinement).
 * Holds the north_star budget on every solve it reports with status 0 -- warm, cold start, interior-point fallback, saturated inputs:
 * <= 1e-4 relative control deviation from the fp64 oracle, teacher-forced (tests/test_gpu_parity.py; observed <= 2.4e-5, median
 * 2e-8 .. 4e-7).  Status 0 is what every solve of the tests reports and all but two of 7 M audited solves of the bench workload (quadrotors
 * that are lost: flagged / MPCQ_SOLVE_NAN); a solve the float factorisation cannot refine says so (see MPCQ_SOLVE_LOW_ACCURACY). */
#define MPCQ_PRECISION_F64 0
#define MPCQ_PRECISION_F32 1

/* Tuning of the box-QP solve (HPIPM's options in the reference's generated solver have no equivalent here: the
 * algorithm differs, the optimum does not -- the QP is strictly convex).  Every field: 0 = the default for the
 * precision; validated by mpcq_create (MPCQ_ERR_INVALID outside the stated range).  The defaults were measured on two
 * workloads; results do not depend on them beyond rounding.  With MPCQ_TUNING=1 in the
 * environment, MPCQ_WARM_MAX, MPCQ_WARM_RETRY, MPCQ_FLIP_MAX, MPCQ_ABORT_PINS, MPCQ_ABORT_WRONG, MPCQ_POLISH_MAX,
 * MPCQ_PIN_RATIO, MPCQ_IPM_MU0, MPCQ_IPM_MARGIN, MPCQ_IPM_TOL, MPCQ_STAGE_MEM=lds|global|compact, MPCQ_GENERIC=1, MPCQ_BLOCK_ORDER,
 * MPCQ_SPLIT_PLANT=0|1 (the plant update between two lockstep periods as its own launch), MPCQ_GROUPS, MPCQ_KEV_STRIDE, MPCQ_VERBOSE=1 override
 * the corresponding field (measurement scripts only; without MPCQ_TUNING=1 the environment is not consulted). */
typedef struct mpcq_tuning {
  int32_t warm_max;     /* passes of the warm active-set attempt, 1..64 (default 12; 6 in fp64 before round 4) */
  int32_t warm_retry;   /* ... in the period after a fallback solve, 1..64 (default 1) */
  int32_t flip_max;     /* changed bound states in a fallback solve above which the next warm attempt is skipped, 1..512; -1: never (default 2) */
  int32_t abort_pins;   /* warm attempt given up when its first pass pins this many inputs, 1..512; -1: never (default 10, N/2 for N > 20) */
  int32_t abort_wrong;  /* ... or a multiplier check finds this many wrong signs, 1..512; -1: never (default 9, 9N/20 for N > 20) */
  int32_t polish_max;   /* active-set passes behind the interior point, 1..64; -1: none, interior point to qp_tol -- MPCQ_PRECISION_F64 only,
                           refused with MPCQ_PRECISION_F32, whose answer comes from these passes (default 16 f64 / 12 f32) */
  int32_t stage_mem;    /* layout of the per-instance working set: 0 automatic, 1 all LDS, 2 per-stage records in global memory (L2),
                           3 compact (since 0.4: Riccati gains in global memory as well, <= 256 registers: more instances per CU) */
  int32_t generic_kernel; /* 1: the any-shape kernel instance even where a shape-specialised one exists */
  double pin_ratio;     /* interior point -> working set: pinned where multiplier > pin_ratio x slack, (0, 1e3] (default 0.2) */
  double ipm_mu0;       /* complementarity of the interior start in units of the gradient scale, [1e-12, 1] (default 1e-4) */
  double ipm_margin;    /* interior start: distance from the bounds in units of their width, (0, 0.5) (default 0.1) */
  double ipm_tol;       /* interior point -> active-set hand-over tolerance, [qp_tol, 1e-1] (default 1e-6 f64 / 1e-5 f32; f32: not below 1e-5).
                         * fp64 with N <= 32: the float interior point in front hands over at a complementarity of 3e-7 (compile-time), this one
                         * is the tolerance of the double interior point that follows a float one that broke down */
  /* ---- since 0.4 */
  int32_t block_order;  /* launch order of a lockstep period: 0 automatic (quadrotors predicted expensive first when the batch exceeds
                           what the device holds at once), 1 never (workgroup p = quadrotor p), 2 always.  Results do not depend on it. */
  /* ---- since 0.6 (0.4 / 0.5: a reserved field that had to be 0 = automatic) */
  int32_t groups;       /* mpcq_sim_steps: the batch as this many contiguous groups, each advancing in lockstep on a HIP stream of its own,
                           1..16; 0 automatic: 1 for a batch that is resident on the device as a whole, 2 for a larger one (the tail of one
                           group's launch -- the device draining while the last workgroups finish -- is filled by the other group's next
                           launch).  A call still ends with EVERY quadrotor K periods on and quadrotors are independent: results do not
                           depend on it.  mpcq_step / mpcq_step_device_async (one period per call) are always one launch over the batch.
                           More than four groups oversubscribe the hardware queues of the device and are slower. */
} mpcq_tuning;

/* Engine configuration.  Replaces the constructor arguments of quad_optimizer
 * (quad, t_horizon, n_nodes, gpe; src/quad_opt.py:36) and the constants it bakes into the
 * generated solver (weights src/quad_opt.py:122-130, bounds :142-144, quad constants
 * src/quad.py:385-417, RGP basis/theta src/gp/RGP.py:126-157). */
typedef struct mpcq_config {
  int32_t batch;      /* B: number of independent quadrotors in this engine (this rank's shard) */
  int32_t N;          /* n_nodes: shooting intervals */
  int32_t nb;         /* RGP basis points per axis (0: nominal model, use_gp=0) */
  int32_t skip;       /* control_freq_factor = int(optimization_dt / 0.01), src/mpc_controller_node.py:222 */
  double T;           /* t_horizon [s] */
  double dt_pred;     /* step of the nominal prediction: ODOMETRY_DT 0.01 (node) or T/N (python sim) */
  double mass, J[3], max_thrust, x_f[4], y_f[4], z_l_tau[4], g;
  double rotor_drag[3], aero_drag; /* plant only (src/quad.py:79-89); unused by the controller path */
  double W[17], W_e[13];           /* diagonal LS weights; stage cost is scaled by T/N (acados) */
  double u_lb[4], u_ub[4], u_ref[4];
  double qp_tol;      /* KKT tolerance of the interior point's last resort; 0 = default for the precision (1e-11 f64, 1e-5 f32: smaller values
                         are raised to 1e-5 there, the f32 answer is refined against fp64 residuals behind the interior point) */
  const double* basis;   /* [3*nb] basis vectors X per axis */
  const double* theta;   /* [3*3] per axis: L, sigma_f, sigma_n */
  int32_t device;        /* HIP device ordinal */
  int32_t precision;     /* MPCQ_PRECISION_* : arithmetic type of the device path */
  int32_t qp_max_iter;   /* 0 = default */
  int32_t flags;         /* MPCQ_FLAG_* */
  double finish_radius;  /* EPSILON_TRAJECTORY_FINISHED [m], src/mpc_controller_node.py:118; 0 = default 1.0 */
  /* ---- since 0.3 (callers built against an older header: mpcq_create_sized with THEIR sizeof(mpcq_config)) */
  mpcq_tuning tune;      /* all-zero = defaults */
} mpcq_config;

typedef struct mpcq_engine mpcq_engine;

const char* mpcq_last_error(void);
/* "mpcq <major.minor> (gfx950, source <16 hex digits>)": the digits are the hash of the sources and the build recipe the library was
 * built from (csrc/Makefile SRC_ID = bench.kernel_source_sha16()); profiles under profiles/ carry the same hash. */
const char* mpcq_version(void);

/* ---- lifetime.  quad_optimizer.__init__ (src/quad_opt.py:36-160): builds constants, K_x^-1,
 * allocates device state and zero-initialises the iterate (acados default), mu=0, C=K_x. */
/* Versioned form: cfg_size = the caller's sizeof(mpcq_config).  Fields behind cfg_size take their defaults (0), so a
 * caller built against an older header keeps working; sizes that end before `device` or exceed this library's struct
 * are refused. */
int mpcq_create_sized(const mpcq_config* cfg, uint64_t cfg_size, mpcq_engine** out);
/* mpcq_create(cfg, out): for source callers an inline that passes THIS header's sizeof(mpcq_config); the exported symbol of the
 * same name exists for binaries built against the 0.3 header only and reads the 0.3 layout (the struct has grown since: a library
 * that copied its own sizeof would read behind such a caller's struct). */
#ifdef MPCQ_BUILDING_LIBRARY
int mpcq_create(const mpcq_config* cfg, mpcq_engine** out);
#else
static inline int mpcq_create(const mpcq_config* cfg, mpcq_engine** out) { return mpcq_create_sized(cfg, sizeof(mpcq_config), out); }
#endif
int mpcq_destroy(mpcq_engine* e);
int mpcq_reset(mpcq_engine* e);

/* ---- reference.  Fused path: whole sampled trajectories stay on the device and the kernel
 * performs get_reference_chunk (src/utils/utils.py:897-931) + set_reference_trajectory
 * (src/quad_opt.py:295-317) itself.  traj [B, Tmax, 13], len [B] (rows valid per instance);
 * resets the trajectory cursor idx_traj to 0 (src/mpc_controller_node.py:517-552). */
int mpcq_set_trajectories(mpcq_engine* e, const double* traj, const int32_t* len, int32_t Tmax);
/* Explicit path: acados .set(j,'yref',·) for j<N and .set(N,'yref',·)  (src/quad_opt.py:311,315).
 * yref [B, N, 17], yrefN [B, 13]. */
int mpcq_set_reference(mpcq_engine* e, const double* yref, const double* yrefN);
/* acados .set(ii,'p',rgp_params) on every stage (src/quad_opt.py:402-404). mu [B, 3*nb] */
int mpcq_set_params(mpcq_engine* e, const double* mu);

/* ---- quad_optimizer.run_optimization (src/quad_opt.py:321-350): pin x0, ONE SQP-RTI iteration
 * on the persisted iterate, using the stored reference and parameters.  x0 [B, 13]. */
int mpcq_solve(mpcq_engine* e, const double* x0);
/* .get(stage,·): one strided device-to-host copy of the B rows of that stage per call */
int mpcq_get_x(mpcq_engine* e, int32_t stage, double* out);      /* .get(stage,'x') -> [B,13] */
int mpcq_get_u(mpcq_engine* e, int32_t stage, double* out);      /* .get(stage,'u') -> [B,4]  */
int mpcq_get_cost(mpcq_engine* e, double* out);                  /* .get_cost()     -> [B]    */
int mpcq_get_status(mpcq_engine* e, int32_t* out);               /* solve() status  -> [B]    */
/* -> [B]: decimal fields of the last solve.  qp_iter % 1000: Riccati factorisations (active-set passes + interior-point
 * iterations); (qp_iter / 1000) % 10 != 0: the warm active-set attempt was given up or skipped, the solve went through
 * the interior point ("fallback solve"); (qp_iter / 10000) % 10 != 0: that solve also moved more than flip_max inputs
 * on/off their bounds (the next solve skips the warm attempt); qp_iter / 100000: why the warm attempt ended
 * (MPCQ_WARM_*), 0 when it succeeded or there was none (cold start). */
int mpcq_get_qp_iter(mpcq_engine* e, int32_t* out);
/* What the last solve of every quadrotor executed, out [B] (read as unsigned): Riccati factorisations (bits 0..14; resumed ones count
 * as whole) | matrix-vector sweeps over the horizon (bits 16..26, saturating) | since 0.6, bits 27..31: how many of the interior-point
 * iterations ran in float (fp64 instances whose interior point iterates in float, N <= 32 on a shape-specialised instance: all of a
 * fallback solve's, unless bit 15 is set; 0 everywhere else -- a latency model prices exactly these with the float chains).  Bit 15 (fp64
 * instances): the float interior point of a fallback solve broke down and the double one ran from the start (a diagnostic: the answer
 * comes from the double active-set method either way).  The dependent chains of these are what a lockstep launch lasts
 * (bench.py `latency_roofline`).  acados reports sqp_iter / qp_iter through get_stats (src/quad_opt.py:337 reads time_tot only). */
int mpcq_get_qp_work(mpcq_engine* e, int32_t* out);
#define MPCQ_WARM_BUDGET 1    /* pass budget (warm_max / warm_retry) exhausted */
#define MPCQ_WARM_PINS 2      /* first pass pinned >= abort_pins inputs */
#define MPCQ_WARM_WRONG 3     /* a multiplier check found >= abort_wrong wrong signs */
#define MPCQ_WARM_BOUNCE 4    /* a bulk release bounced back with most inputs saturated */
#define MPCQ_WARM_NUMERIC 5   /* a stage Hessian was not positive definite / not a number */
#define MPCQ_WARM_SKIPPED 6   /* skipped: the previous solve carried the flip mark */
/* .get_stats('time_tot'): device time of the last solve/step launch in seconds (whole batch) */
int mpcq_get_stats(mpcq_engine* e, double* time_tot);

/* ---- quad_optimizer.discrete_dynamics on the nominal model (src/quad_opt.py:353-377,
 * src/mpc_controller_node.py:298).  x [B,13], u [B,4] -> out [B,13] */
int mpcq_predict_nominal(mpcq_engine* e, const double* x, const double* u, double dt, double* out);

/* ---- quad_optimizer.regress_and_update_RGP_model (src/quad_opt.py:380-406): 3 scalar RGP
 * Kalman updates per instance and the new means become the stage parameters.
 * v_body [B,3], a_drag [B,3]. */
int mpcq_rgp_regress(mpcq_engine* e, const double* v_body, const double* a_drag);
int mpcq_get_rgp(mpcq_engine* e, double* mu /*[B,3,nb] or NULL*/, double* C /*[B,3,nb,nb] or NULL*/);

/* ---- fused control step = the loop body src/mpc_controller_node.py:278-318
 * (src/execute_trajectory.py:202-258): chunk -> yref -> solve -> w=U[0] -> nominal prediction ->
 * idx_traj++ -> compute_a_drag -> RGP regress -> params.  x_meas [B,13] -> w_out [B,4];
 * x_pred_out [B,13] may be NULL. */
int mpcq_step(mpcq_engine* e, const double* x_meas, double* w_out, double* x_pred_out);
/* Same with device-resident buffers: d_x_meas [B,13] and d_w_out [B,4] (NULL: engine-internal) are
 * float64 device pointers in EVERY precision (the measurement and the iterate are always double; the
 * precision only selects the arithmetic of the QP).  No host traffic, no synchronisation; ordered on
 * the engine's stream. */
int mpcq_step_device_async(mpcq_engine* e, const double* d_x_meas, double* d_w_out);
int mpcq_synchronize(mpcq_engine* e);
void* mpcq_stream(mpcq_engine* e);   /* hipStream_t the engine launches on */

/* ---- command mapping of publish_control_gazebo (src/mpc_controller_node.py:600-612) for the last
 * step / solve: rotor_thrusts [B,4] = w * max_thrust / mass, collective_thrust [B] = sum(w) * max_thrust
 * / mass, bodyrates [B,3] = x_opt[1, 10:13] (src/mpc_controller_node.py:292).  Any pointer may be NULL. */
int mpcq_get_command(mpcq_engine* e, double* rotor_thrusts, double* collective_thrust, double* bodyrates);
/* ---- trajectory finished (src/mpc_controller_node.py:374): per instance 1 once a fused step found
 * idx_traj + 1 == len(trajectory) (after its idx_traj += 1) with the quadrotor closer than
 * finish_radius to the first row of that step's reference chunk; sticky until mpcq_set_trajectories
 * or mpcq_reset.  out [B]. */
int mpcq_get_finished(mpcq_engine* e, int32_t* out);
/* get_reference_chunk (src/utils/utils.py:897-931) at the current cursor, evaluated on the device
 * with the row selection of the fused step.  out [B, N, 13]. */
int mpcq_get_reference_chunk(mpcq_engine* e, double* out);

/* ---- closed-loop harness on the device: Quadrotor3D.update with drag
 * (src/quad.py:166-190,234-277,329-357) applied n_sub times with step sim_dt to the engine's
 * internal plant state, driven by the last w.  mpcq_sim_reset sets the plant state [B,13];
 * mpcq_sim_steps runs K closed-loop iterations {step(x) -> plant} without host round trips. */
int mpcq_sim_reset(mpcq_engine* e, const double* x0);
int mpcq_sim_steps(mpcq_engine* e, int32_t K, int32_t n_sub, double sim_dt);
/* The same K closed-loop iterations as ONE launch in which every instance runs through its K control periods
 * without waiting for the others (instances are independent; src/execute_trajectory.py:196-279 is a loop over ONE
 * quadrotor).  Same arithmetic, same results as mpcq_sim_steps; the per-period outputs readable afterwards
 * (mpcq_get_*, mpcq_sim_get_state) are those of the last period. */
int mpcq_sim_run(mpcq_engine* e, int32_t K, int32_t n_sub, double sim_dt);
/* The reference's plant loop between two solves (src/execute_trajectory.py:232-243):
 * `while control_time < optimization_dt: quad.update(w, simulation_dt); control_time += simulation_dt`.
 * mpcq_plant_substeps reproduces its iteration count with the same double accumulation (20 / 11 / 4
 * substeps for control_dt = 0.1 / 0.05 / 0.02 at sim_dt = 5e-3); negative on bad arguments.
 * mpcq_sim_plant_period advances the plant state by that loop with control w [B,4] (NULL: the last
 * control of the engine); mpcq_sim_control_periods = K x {fused step -> that loop}. */
int mpcq_plant_substeps(double control_dt, double sim_dt);
int mpcq_sim_plant_period(mpcq_engine* e, const double* w, double control_dt, double sim_dt, int32_t* n_sub /*out, may be NULL*/);
int mpcq_sim_control_periods(mpcq_engine* e, int32_t K, double control_dt, double sim_dt, int32_t* n_sub /*out, may be NULL*/);
int mpcq_sim_get_state(mpcq_engine* e, double* x /*[B,13]*/, double* w /*[B,4] or NULL*/);
/* HIP-event time of the step-kernel launches of the last mpcq_sim_steps / mpcq_sim_run call (events recorded on
 * the engine's stream around every 4th launch, MPCQ_KEV_STRIDE=1 for every launch): total seconds of the timed
 * launches and their number (mpcq_tuning.groups > 1: the launches of group 0, see mpcq_get_groups). */
int mpcq_get_kernel_time(mpcq_engine* e, double* seconds, int32_t* launches);
int mpcq_get_kernel_time_minmax(mpcq_engine* e, double* fastest_s, double* slowest_s);   /* of the same timed launches */
/* diagnostic build only (libmpcq_prof.so, -DMPCQ_PROFILE): per-instance shader-cycle totals per phase of
 * the last step, out [B][16]; MPCQ_ERR_STATE in the product build. */
int mpcq_debug_profile(mpcq_engine* e, unsigned long long* out);
/* The launch order of the last lockstep period, out [B]: workgroup p ran quadrotor out[p] (mpcq_tuning.block_order; the
 * identity when no order is in use).  The reference solves one quadrotor per process (src/quad_opt.py:321-350) and has no
 * counterpart; results do not depend on the order. */
int mpcq_get_block_order(mpcq_engine* e, int32_t* out);
/* Since 0.6.  The number of groups mpcq_sim_steps runs this engine's batch in (mpcq_tuning.groups resolved: 1 = one launch per period
 * over the whole batch).  With more than one group the launches mpcq_get_kernel_time reports are group 0's -- a launch over B / groups
 * quadrotors that shares the device with the other groups' launches: per-period figures come from the caller's clock around the call.
 * No counterpart in the reference (one quadrotor per process, src/quad_opt.py:321-350). */
int mpcq_get_groups(mpcq_engine* e, int32_t* out);

/* ---- tracking statistic (src/Visualiser.py:787-789,809-811,918), summed over this engine's
 * instances since the last reset: out[0]=sum |e_pos|^2, out[1]=sum |e_vel|^2, out[2]=steps,
 * out[3]=max |e_pos|^2, out[4]=instances with status != 0 in the last step. */
int mpcq_get_tracking_stats(mpcq_engine* e, double out[5]);

/* ---- multi-GPU: one engine per rank; the only collective is the reduction of the statistics
 * vector (RCCL over xGMI).  The host exchanges the 128-byte unique id however it likes. */
int mpcq_comm_unique_id(void* id128);
int mpcq_comm_init(mpcq_engine* e, int32_t rank, int32_t nranks, const void* id128);
/* A second engine of the same process and device reduces over the communicator `owner` initialised (since 0.5): one rank = one
 * communicator however many engines it runs (bench.py: the configs[3] swarm next to the configs[1] headline).  `e` borrows the
 * handle, `owner` has to outlive it (or `e` must not reduce any more); calls on the two engines must not overlap in time.
 * The RCCL library is dlopen'ed by name (librccl.so, then /opt/rocm/lib/librccl.so); MPCQ_RCCL_LIB in the environment names another file. */
int mpcq_comm_share(mpcq_engine* e, mpcq_engine* owner);
/* sum (out[0..2], out[4]) / max (out[3]) over ranks; every rank receives the result */
int mpcq_allreduce_tracking_stats(mpcq_engine* e, double out[5]);

/* ---- state dump / restore (teacher-forced parity tests, checkpoint/resume).  Any pointer may
 * be NULL.  X [B,N+1,13], U [B,N,4], mu [B,3,nb], C [B,3,nb,nb], x_pred_prev [B,13],
 * has_prev [B], idx [B]. */
int mpcq_get_state(mpcq_engine* e, double* X, double* U, double* mu, double* C, double* x_pred_prev,
                   int32_t* has_prev, int32_t* idx);
int mpcq_set_state(mpcq_engine* e, const double* X, const double* U, const double* mu, const double* C,
                   const double* x_pred_prev, const int32_t* has_prev, const int32_t* idx);

/* The rest of the resumable state: warm-start flag / pass count of the last solve qp_iter [B], the
 * tracking accumulators stats [B,4] (sum |e_pos|^2, sum |e_vel|^2, steps, max |e_pos|^2) and the
 * finished flags [B].  With mpcq_get_state + mpcq_sim_get_state a restored engine continues bit for bit. */
int mpcq_get_solver_state(mpcq_engine* e, int32_t* qp_iter, double* stats, int32_t* finished);
int mpcq_set_solver_state(mpcq_engine* e, const int32_t* qp_iter, const double* stats, const int32_t* finished);

/* ---- RGP.learn (src/gp/RGP.py:332-505), SURVEY §8 f4: hyper-parameter learning of the recursive GP (unscented
 * transform over eta = (L, sigma_f, sigma_n) + Kalman / smoother updates) for batch x 3 independent (quadrotor, axis)
 * regressors on the device, fp64.  The loop body never calls learn in the reference (offline estimator), so this is an
 * object of its own: basis [3, nb] and theta [3, 3] as in mpcq_config (initial values, shared by the batch), nb <= 64.
 * mpcq_learn_step feeds one sample per regressor: v_body [B,3] (inputs), a_drag [B,3] (targets).  mpcq_learn_get:
 * mu_g [B,3,nb], C_g [B,3,nb,nb], mu_eta [B,3,3], C_eta [B,3,3,3], Kx_inv [B,3,nb,nb] (K_x^-1 rebuilt for the new
 * hyper-parameters, RGP.py:499-500); any pointer may be NULL.  Errors: mpcq_learn_last_error(). */
typedef struct mpcq_learner mpcq_learner;
const char* mpcq_learn_last_error(void);
int mpcq_learn_create(int32_t batch, int32_t nb, const double* basis, const double* theta, int32_t device, mpcq_learner** out);
int mpcq_learn_destroy(mpcq_learner* l);
int mpcq_learn_step(mpcq_learner* l, const double* v_body, const double* a_drag);
int mpcq_learn_get(mpcq_learner* l, double* mu_g, double* C_g, double* mu_eta, double* C_eta, double* Kx_inv);

#ifdef __cplusplus
}
#endif
#endif /* MPCQ_H */
