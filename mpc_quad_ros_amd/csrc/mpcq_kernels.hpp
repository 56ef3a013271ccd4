// mpcq_kernels.hpp — device code of the batched MPC+RGP control step (gfx950 / CDNA4).
//
// One quadrotor per 64-lane wavefront (one workgroup = one wave).  The per-instance working set lives in
// LDS (iterate, Riccati gains, QP vectors, RGP covariance, shooting records); the per-stage records
// (sensitivities AB'', shooting gaps, cost gradients) live in LDS or, when that lets the whole batch be
// resident at once, in a per-instance global record streamed through L2 (Cfg::GAB).  HBM is touched once
// per step to load and once to store the persistent state, with lane-contiguous (coalesced) records.
// The Riccati factorisation runs its matrix-matrix products on the matrix cores (the recursion vector rides in pad column 14
// of the tiles); the matrix-vector sweeps run on the vector ALU with permlane / DPP cross-lane traffic, without LDS round
// trips or barriers on their critical path.
//
// Precision: the SQP iterate (X, U), the measurement, the reference and every difference that
// defines the QP data (x0 - X0, X_i - xref_i, U_i - uref_i, bounds, shooting gaps) are formed in
// double.  TQ (float or double) is the arithmetic of the model evaluation, the sensitivities and
// the QP solve.  TQ = double reproduces the fp64 oracle to ~1e-10; TQ = float is the fast path.
//
// Algorithm (same mathematical step as the reference's acados SQP-RTI call, restated in
// SURVEY App. A; not a translation of acados/HPIPM code):
//   1. multiple shooting: explicit RK4 (1 step) with forward sensitivities per interval
//   2. box-QP in du on the stage-sparse problem (O(N) memory; the condensed Hessian is never formed):
//      warm active-set method from the previous step's working set -- per working set ONE masked Riccati
//      factorisation + ONE forward sweep of the affine LQ problem (polish); cold start / cycling: Mehrotra
//      predictor-corrector IPM to a hand-over tolerance, then the same active-set iterations to an exact KKT point
//   3. full step, cost, nominal RK4 prediction, body-frame drag estimate, 3 scalar RGP updates.
// Cfg::RUN instances iterate steps 1-3 plus the drag plant over many control periods in one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#ifdef MPCQ_EMU_DEBUG
#include <cstdio>
#endif

namespace mpcq {

// unroll factors of the stage loops (factorisation / vector sweeps); the shape-specialised translation unit sets its own
#ifndef MPCQ_UNROLL_FACTOR
#define MPCQ_UNROLL_FACTOR 1
#endif
#ifndef MPCQ_UNROLL_SWEEP
#define MPCQ_UNROLL_SWEEP 1
#endif
// The four Runge-Kutta substages of the shooting / prediction / plant integrators as ONE loop body (rolled) instead of four
// inlined copies, and the substage loop of the sensitivity pass likewise: the step executes every phase once per control
// period, so straight-line code is fetched from L2 every time (18 % of the wave cycles waited for instructions with
// everything unrolled); a loop body is fetched once and then runs from the instruction cache.
#ifndef MPCQ_ROLL_RK
#define MPCQ_ROLL_RK 1
#endif
// Round-6 restructurings of the glue between the phases: built, measured on the MI355X (tools/r6_ab.sh, tools/r6_ab4.sh,
// profiles/r6_glue_ab.txt) and NOT in the product -- the step kernel sits at the edge of the register file (256 VGPRs, 160+ SGPR spills),
// and what an edit saves in instructions the allocation it shifts gives back:
//   MPCQ_G_SHFL = 1     GP sums of a stage's three lanes through ds_bpermute instead of LDS memory + two barriers      0 .. - 2 %
//   MPCQ_G_DXBATCH = 1  gap operand of a working-set pass: the global loads of a block issued together                 0 .. - 2 %
//   (tools/experiments/r6_glue_restructurings.patch:)  the new iterate kept in registers between the full step and the cost: - 7 %
//   (20 KB more code, SGPR spills 190 -> 340); the compact fp64 layout without X in LDS -- 22.1 instead of 24.3 KB, SEVEN quadrotors
//   per CU: + 0.5 %, a seventh resident wavefront buys nothing once the groups fill the launch tails; fused ratio-test / step passes: 0;
//   the solver's previous-period record fetched with the load phase instead of in front of the QP: - 1 %.
#ifndef MPCQ_G_SHFL
#define MPCQ_G_SHFL 0
#endif
#ifndef MPCQ_G_DXBATCH
#define MPCQ_G_DXBATCH 0
#endif
#if MPCQ_ROLL_RK
#define MPCQ_RK_LOOP _Pragma("clang loop unroll(disable)")
#else
#define MPCQ_RK_LOOP _Pragma("unroll")
#endif
#ifndef MPCQ_ROLL_SENS
#define MPCQ_ROLL_SENS 1
#endif
#if MPCQ_ROLL_SENS
#define MPCQ_SENS_LOOP _Pragma("clang loop unroll(disable)")
#else
#define MPCQ_SENS_LOOP _Pragma("unroll")
#endif
// Phases of the step: inlined into the kernel by default; -DMPCQ_NOINLINE_PHASES keeps them as functions so that
// tools/kernel_resources.sh can attribute the code size (measurement only).  MPCQ_COLD marks the phases only the
// interior-point fallback uses: out of line, so that the many call sites of the fallback do not each carry a copy.
#ifdef MPCQ_NOINLINE_PHASES
#define MPCQ_PHASE __device__ __attribute__((noinline))
#else
#define MPCQ_PHASE __device__ inline
#endif
#ifndef MPCQ_COLD
#define MPCQ_COLD __device__ inline
#endif
template <typename T> struct alignas(16) V4 { T a, b, c, d; };
constexpr int NX = 13, NU = 4, NY = 17;

extern __shared__ unsigned char smem_raw[];   // dynamic LDS (16-byte aligned base), shared by the kernels and their out-of-line phases

// ------------------------------------------------------------------ checked build (-DMPCQ_CHECKED, libmpcq_checked.so)
// Diagnostic build for the GPU (AddressSanitizer is not available on the device here): every pointer of the step kernel
// is a fat pointer P<T> = (base, valid index range, region tag); every access checks its index against the region -- the LDS
// workspace [0, qtotal), the double block [0, dbytes/8), the per-instance global record [0, gtotal), the trajectory
// [0, Tmax 13), the state records -- and every cross-lane operation (v_readlane, DPP, permlane swaps, MFMA) checks that
// the wavefront executes it with a full EXEC mask.  The first violation is recorded in DevState::chk (tag, index, range,
// lane, workgroup, program counter relative to the kernel entry) and the access is redirected to a sink, so the launch
// completes and the host reports it (mpcq_api.hip: every launch of the checked build is followed by a read of the record).
// In the product build P<T> is T* and none of this exists.
enum : int { CK_LDS_D = 1, CK_LDS_Q, CK_STAGE, CK_X, CK_U, CK_TRAJ, CK_YREF, CK_YREFN, CK_MU, CK_C, CK_XPP, CK_W, CK_XPRED, CK_STATS, CK_RUNX,
             CK_XMEAS, CK_BASIS, CK_KXINV, CK_NULL, CK_EXEC = 64 };
#ifdef MPCQ_CHECKED
constexpr int CK_HDR = 16;   // bytes at the base of the dynamic LDS block: the address of the violation record
__device__ inline int* ck_rec() { return *reinterpret_cast<int**>(smem_raw); }
__device__ inline void ck_report(int tag, long i, long lo, long hi) {
  int* r = ck_rec();
  if (r && atomicCAS(&r[0], 0, 1) == 0) {
    r[1] = tag; r[2] = (int)i; r[3] = (int)lo; r[4] = (int)hi; r[5] = (int)threadIdx.x; r[6] = (int)blockIdx.x;
    const unsigned long long pc = __builtin_amdgcn_s_getpc();
    r[7] = (int)(unsigned)pc; r[8] = (int)(unsigned)(pc >> 32);
  }
}
__device__ inline void ck_exec(int site) {
  const unsigned long long ex = __builtin_amdgcn_read_exec();
  if (ex != ~0ull) ck_report(CK_EXEC + site, (long)(ex >> 32), 0, (long)(unsigned)ex);
}
template <typename T> struct CkPtr {
  using V = typename std::remove_const<T>::type;
  T* p; long lo, hi; int tag;
  mutable V sink;
  __device__ CkPtr() : p(nullptr), lo(0), hi(0), tag(CK_NULL), sink() {}
  __device__ CkPtr(decltype(nullptr)) : CkPtr() {}
  __device__ CkPtr(T* p_, long n, int tag_) : p(p_), lo(0), hi(p_ ? n : 0), tag(tag_), sink() {}
  __device__ CkPtr(T* p_, long lo_, long hi_, int tag_) : p(p_), lo(lo_), hi(hi_), tag(tag_), sink() {}
  template <typename U> __device__ CkPtr(const CkPtr<U>& o) : p(o.p), lo(o.lo), hi(o.hi), tag(o.tag), sink() {}
  __device__ explicit operator bool() const { return p != nullptr; }
  // element i (and the n - 1 behind it) inside the region?
  __device__ bool ok(long i, long n = 1) const {
    if (i >= lo && i + n <= hi) return true;
    ck_report(tag, i, lo, hi);
    return false;
  }
  template <typename I> __device__ T& operator[](I i) const { return ok((long)i) ? p[(long)i] : const_cast<T&>(static_cast<const V&>(sink)); }
  template <typename I> __device__ CkPtr operator+(I o) const { return CkPtr(p + (long)o, lo - (long)o, hi - (long)o, tag); }
  __device__ T& c(long k) const { return (*this)[k]; }   // see OffPtr::c
};
template <typename T> using P = CkPtr<T>;
template <typename T> __device__ inline P<T> mk(T* p, long n, int tag) { return P<T>(p, n, tag); }
// 128-bit view of four consecutive elements
template <typename T> __device__ inline V4<typename std::remove_const<T>::type> ld4(const P<T>& b, long off) {
  if (b.ok(off, 4)) return *reinterpret_cast<const V4<typename std::remove_const<T>::type>*>(b.p + off);
  return V4<typename std::remove_const<T>::type>{};
}
template <typename T> __device__ inline void st4(const P<T>& b, long off, const V4<T>& v) {
  if (b.ok(off, 4)) *reinterpret_cast<V4<T>*>(b.p + off) = v;
}
// float view of a double region (the float interior point of the fp64 instances works in the space of the double QP vectors)
__device__ inline P<float> as_float(const P<double>& d) { return P<float>(reinterpret_cast<float*>(d.p), 2 * d.lo, 2 * d.hi, d.tag); }
#define CK_EXEC_FULL(site) ck_exec(site)
#else
constexpr int CK_HDR = 0;
// Product build: P<T> = base pointer + unsigned 32-bit BYTE offset, kept apart until the access.  The base of every region is
// wave-uniform (the LDS workspace, the record of this workgroup's quadrotor), so an access to a global record becomes
// `global_load/store v, v_offset, s[base:base+1]` -- one 32-bit add per address where plain pointer arithmetic on T* costs an index
// add, a sign extension and a 64-bit shift-add (three vector instructions per load; a fifth of the forward sweep over global stage
// records was address arithmetic).  Offsets wrap modulo 2^32: every region is far smaller than 4 GiB and no access lies in front of
// its region's base (the checked build verifies both).
template <typename T> struct OffPtr {
  using V = typename std::remove_const<T>::type;
  T* b; unsigned o;
  __host__ __device__ OffPtr() : b(nullptr), o(0) {}
  __host__ __device__ OffPtr(decltype(nullptr)) : b(nullptr), o(0) {}
  __host__ __device__ OffPtr(T* p) : b(p), o(0) {}
  __host__ __device__ OffPtr(T* p, unsigned o_) : b(p), o(o_) {}
  template <typename U> __host__ __device__ OffPtr(const OffPtr<U>& x) : b(x.b), o(x.o) {}
  __host__ __device__ explicit operator bool() const { return b != nullptr; }
  template <typename I> __host__ __device__ T& operator[](I i) const {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<V*>(b)) + (o + (unsigned)i * (unsigned)sizeof(T)));
  }
  template <typename I> __host__ __device__ OffPtr operator+(I i) const { return OffPtr(b, o + (unsigned)i * (unsigned)sizeof(T)); }
  // element at a compile-time-constant distance k (after unrolling): added behind the zero extension, where it becomes the
  // immediate offset field of the instruction -- (p + lane_part).c(k) for several k shares one address register
  __host__ __device__ T& c(long k) const { return *reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<V*>(b)) + (size_t)o + k * (long)sizeof(T)); }
};
template <typename T> using P = OffPtr<T>;
template <typename T> __device__ inline P<T> mk(T* p, long, int) { return P<T>(p); }
template <typename T> __device__ inline V4<typename std::remove_const<T>::type> ld4(const P<T>& b, long off) {
  return *reinterpret_cast<const V4<typename std::remove_const<T>::type>*>(&b[off]);
}
template <typename T> __device__ inline void st4(const P<T>& b, long off, const V4<T>& v) { *reinterpret_cast<V4<T>*>(&b[off]) = v; }
// float view of a double region (the float interior point of the fp64 instances works in the space of the double QP vectors)
__device__ inline P<float> as_float(const P<double>& d) { return P<float>(reinterpret_cast<float*>(d.b), d.o); }
#define CK_EXEC_FULL(site)
#endif
constexpr int ABW = 16;          // row stride of AB'' = [A[:, q v r] | B]: the columns of [A|B] that are not [0;I] (position)
constexpr int VS = 16;           // stride of state-sized QP vectors (internal order, 13 used)
constexpr int ABS = NX * ABW;    // per-stage stride of AB'
constexpr int PST = 256 + VS;     // stride of a stored cost-to-go (see Lds::pst)
// Cost-to-go tiles are kept for the stages i = 1 (mod PSTEP) only: a factorisation that has to restart below a changed stage t restarts at
// the next multiple of PSTEP at or above t (from the tile of the stage above it), i.e. revisits at most PSTEP - 1 stages more than the
// change requires -- and writes a quarter of the tiles (2.2 KB each in fp64; at N = 20 the tiles were 43 KB of the ~100 KB a quadrotor
// wrote per period).  A restart from an exact copy of the tile reproduces the full factorisation bit for bit, whatever the spacing.
constexpr int PSTEP = 4;
__host__ __device__ inline int pst_first(int N, int start, int pst_valid) {   // stage a factorisation (re)starts at; N - 1: from the top
  if (start < 0) return N - 1;
  const int f = (start + PSTEP - 1) & ~(PSTEP - 1);
  return (f < N - 1 && f + 1 <= pst_valid) ? f : N - 1;
}
constexpr int MROW = 20;          // stride of a multiplier row (see Lds::mrow)
constexpr int KS = NU * ABW;     // per-stage stride of K (4 rows of 13, padded to 16)
// per (stage, RK substage) record of the shooting pass, what the sensitivity pass reads: q_s(4) r_s(3) | d vdot/dq (3x4) | d vdot/dv (3x3) | R[:,2]
constexpr int SUB_Q = 0, SUB_R = 4, SUB_JQ = 7, SUB_JV = 19, SUB_RZ = 28, SUBW = 31;
constexpr int SUBS = 4 * SUBW + 1;  // per-stage stride of the records (odd: lanes of different stages hit different banks)

enum : int { MODE_TRAJ = 1, MODE_POST = 2, MODE_RUN = 4, MODE_PLANT_FIRST = 8, MODE_STATIC_GP = 16 };
// why a warm active-set attempt ended without a solution (field x 100000 of qp_iter; 0: it was not given up / there was none)
enum : int { QPX_BUDGET = 1, QPX_PINS = 2, QPX_WRONG = 3, QPX_BOUNCE = 4, QPX_NUMERIC = 5, QPX_SKIPPED = 6 };
// MODE_STATIC_GP: the GP of the model is fixed (mpcq_config.flags & MPCQ_FLAG_STATIC_GP): the post phase skips the RGP update.
// MODE_RUN: free-running closed loop, DevState::run_* periods per launch.  MODE_PLANT_FIRST: the launch starts by
// advancing the plant state run_x with the previous launch's control (lockstep closed loop without a plant kernel
// between two step launches: the integration overlaps the global loads of the load phase).

// Diagnostic build only (-DMPCQ_PROFILE, libmpcq_prof.so): per-phase shader-cycle totals per instance.
enum : int { PF_LOAD = 0, PF_SHOOT_X, PF_SHOOT_S, PF_FACTOR, PF_FWD, PF_BWD, PF_ADJ, PF_ROLL, PF_ELEM, PF_POST, PF_TOTAL, PF_N = 16 };
#ifdef MPCQ_PROFILE
struct Prof { unsigned long long acc[PF_N]; unsigned long long t; int oth; };
__device__ inline void pf_start(Prof& p) { p.t = __builtin_readcyclecounter(); }
__device__ inline void pf_stop(Prof& p, int k) { const unsigned long long n = __builtin_readcyclecounter(); p.acc[k] += n - p.t; p.t = n; }
#define PF_ARG , Prof& pf
#define PF_PASS , pf
#ifdef MPCQ_PROFILE_NOSTAMP
#define PF_START()
#define PF_STOP(k)
#elif defined(MPCQ_PROFILE_OTHER)   // what lies BETWEEN the bracketed phases of the fp64 step, by bucket (slots 11..15; tools/profile_phases.py PROF_MODE=other)
#define PF_START() pf_stop(pf, pf.oth)
#define PF_STOP(k) pf_stop(pf, k)
#else
#define PF_START() pf_start(pf)
#define PF_STOP(k) pf_stop(pf, k)
#endif
#else
#define PF_ARG
#define PF_PASS
#define PF_START()
#define PF_STOP(k)
#endif
#if defined(MPCQ_PROFILE) && defined(MPCQ_PROFILE_OTHER)
#define PF_MARK(k) do { pf_stop(pf, pf.oth); pf.oth = (k); } while (0)   // close the running bucket, open bucket k
#define PF_BUCKET(k) pf.oth = (k)
#else
#define PF_MARK(k)
#define PF_BUCKET(k)
#endif
#if defined(MPCQ_PROFILE) && defined(MPCQ_PROFILE_FWD)
#define PF_FINE(k) pf_stop(pf, k)
#else
#define PF_FINE(k)
#endif
#if defined(MPCQ_PROFILE) && defined(MPCQ_PROFILE_SERIAL)   // cycle stamps around the single-lane blocks and the parts of the post phase (slots 11..15)
#define PF_SER(k) pf_stop(pf, k)
#else
#define PF_SER(k)
#endif
#if defined(MPCQ_PROFILE) && defined(MPCQ_PROFILE_FAC)   // cycle stamps inside a factorisation stage (slots 11..15)
#define PF_FAC(k) pf_stop(pf, k)
#else
#define PF_FAC(k)
#endif

template <typename TQ>
struct DevModel {
  int N, nb, skip, Tmax, B, qp_max_iter, polish_max, warm_max;
  int warm_retry; // cap of the warm attempt in the period after one that fell back to the interior point
  int flip_max;   // more changed bound states than this in a fallback solve: the next period skips the warm attempt (< 0: never)
  int abort_pins; // warm attempt given up after its first pass when that pins at least this many inputs (0: never)
  int abort_wrong; // ... or when a multiplier check finds at least this many wrong signs (0: never)
  int gab;   // stage records (AB'', c, qv) live in DevState::stage instead of LDS (must match the kernel instantiation)
  double h, dt_pred;
  double mass, J[3], tmax, xf[4], yf[4], zl[4], g;
  double imass, iJ[3];   // reciprocals formed once on the host: no divisions in the model evaluations
  double W[NY], We[NX], ulb[NU], uub[NU], uref[NU];
  double rotor_drag[3], aero_drag;
  double finish_r;   // EPSILON_TRAJECTORY_FINISHED (src/mpc_controller_node.py:118)
  TQ qp_tol;    // final KKT tolerance (IPM-only fallback)
  TQ ipm_tol;   // IPM -> active-set polish hand-over tolerance
  TQ ipm_margin;   // interior start: distance from the bounds in units of their width
  TQ pin_ratio;    // working set taken over from the interior point: input pinned where multiplier > pin_ratio x slack
  TQ ipm_mu0;   // initial complementarity of the interior start, in units of the gradient scale
  TQ eps;       // unit roundoff scale of TQ used for KKT sign / bound tests
  TQ L2inv[3], sf2[3], sn2[3];
  const TQ* basis;  // [3*nb]
  const TQ* Kxinv;  // [3*nb*nb]
#ifdef MPCQ_DUMP_AT   // reproducer builds only (tools/repro_codegen): [B][4096] doubles, what the chosen point of the first interior-point iteration holds
  double* dbg;
#endif
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
  double* w_ext;    // [B][4] caller's control buffer of mpcq_step_device_async (nullptr: none); st.w is written as well
  double* xpred;    // [B][13]
  double* cost;     // [B]
  double* stats;    // [B][4]
  int* has_prev;
  int* idx;
  const int* tlen;
  int* status;
  int* qp_iter;
  int* qp_work;     // [B] what the last solve executed: factorisations (bits 0..14) | float interior point broke down (bit 15) | vector sweeps (bits 16..26) | interior-point iterations run in float (bits 27..31) (mpcq_get_qp_work)
  int* finished;    // [B] trajectory finished (src/mpc_controller_node.py:374), sticky until new trajectories / reset
  TQ* stage;        // [B][Lds::gtotal] per-instance stage records (GAB layouts only)
  double* run_x;    // [B][13] plant states of the free-running closed loop (MODE_RUN; aliases x_meas)
  int run_steps, run_nsub;   // control periods per launch, plant substeps per period
  double run_dt;    // plant substep
  unsigned long long* prof;   // [B][PF_N] (diagnostic build only)
  int* chk;         // [16] first violation found by the checked build (-DMPCQ_CHECKED), nullptr otherwise
  const int* order; // workgroup p runs quadrotor order[p] (order_kernel: expensive quadrotors first; global indices); nullptr: b0 + p
  int b0;           // first quadrotor of this launch: a launch covers the contiguous group [b0, b0 + gridDim.x) of the batch (mpcq_sim_steps with
                    // tune.groups > 1 runs the groups' periods on streams of their own; `order` then points at the group's segment)
};

// ------------------------------------------------------------------ LDS layout
// doubles first (offsets in doubles from the LDS base), then the TQ region (offsets in TQ elements
// from the TQ base = base + dbytes).  State-sized QP vectors use the INTERNAL state order
// [q(4) v(3) r(3) p(3)] (position last) and a stride of 16.
struct Lds {
  int X, U, x0, pre, dbytes;
  int zd, dxd;               // mixed precision (TQ = float): the QP solution and its state trajectory in double (refined against fp64 residuals)
  int curv;                  // mixed precision: (B' P B)_aa of every input from the last working-set factorisation (what releasing a pinned input would move it by: polish_mixed)
  int AB, c, qv, r0, lb, ub, alpha, basis, wq;
  int z, sl, su, ll, lu, grad, vin, dza, dz, rho, act, rt, dx, Dx, K, Linv, sF, sT, stv;
  int sub, rgp, qtotal;
  int gab, zb, gx, gtotal;   // stage data (AB'', c, qv) in global memory? ; LDS zero block ; GP exchange scratch (MPCQ_G_SHFL = 0 only) ; global elements per instance
  int gk;                    // Riccati gains K, Lambda^-1 in the global record as well, and r0 / lb / ub inside the union (written behind the shooting)
  int mrow;                  // multiplier rows (always global): per stage 4 rows [M_a(13) | F_uu row(4) | gt_a | pad 2]
  int pst;                   // cost-to-go of every stage (always global): [P_i as accumulator tile (256) | p_i (16)]
};
__host__ __device__ inline int al4(int v) { return (v + 3) & ~3; }
// gab: 0 everything in LDS | 1 the stage records (AB'', c, qv) in the per-instance global record | 2 = 1 + the Riccati gains there too
// and the QP vectors r0 / lb / ub inside the union region (the "compact" layout of large batches: six instead of four quadrotors
// per CU in fp64 at N = 20, i.e. a second wave on two of the four SIMDs).
// mixed: the instance computes in float and refines its QP solution against residuals evaluated in double (TQ = float)
__host__ __device__ inline Lds lds_layout(int N, int nb, int gab, int mixed = 0) {
  Lds L;
  int o = 0, g = 0;
  L.gk = gab == 2 ? 1 : 0;
  gab = gab ? 1 : 0;
  auto take = [&](int n) { int r = o; o += al4(n); return r; };
  auto gtake = [&](int n) { int r = g; g += al4(n); return r; };
  L.X = take((N + 1) * NX);
  L.U = take(N * NU);
  L.x0 = take(NX + 8);   // + [v_body(3), a_drag(3)] scratch of the post phase
  L.pre = take(NX + 5);  // what the post phase reads of the persistent state, fetched in the load phase: x_pred_prev(13) | statistics(4) | has_prev
  L.zd = L.dxd = 0;
  if (mixed) L.zd = take(N * NU);   // (L.dxd: set below, it lives in the space of the float vectors dx | Dx)
  L.dbytes = o * 8;
  o = 0;
  const int nv = N * NU;
  L.gab = gab;
  if (gab) {   // per-stage data streamed from global memory (L2 / MALL): LDS keeps only the QP workspace
    L.AB = gtake(N * ABS + VS);   // + one zero block read by padding lanes
    L.c = gtake(N * VS);
    L.qv = gtake((N + 1) * VS);
    L.zb = take(VS);
    L.gx = MPCQ_G_SHFL ? 0 : take(22 * 8);   // 21 lane triples + the idle lane 63
  } else {
    L.AB = take(N * ABS + VS);
    L.c = take(N * VS);
    L.qv = take((N + 1) * VS);
    L.zb = L.AB + N * ABS;
    L.gx = L.AB;   // (MPCQ_G_SHFL = 0: exchange scratch of shoot_states; AB'' is not written before shoot_sens)
  }
  L.mrow = gtake(N * MROW * NU);
  L.pst = gtake(((N + PSTEP - 1) / PSTEP) * PST);   // one tile per PSTEP stages (pst_first)
  if (!L.gk) { L.r0 = take(nv); L.lb = take(nv); L.ub = take(nv); }
  L.alpha = take(3 * nb);
  L.basis = take(3 * nb);
  L.wq = take(3 * VS);   // stage / terminal state weights in internal order, input weights
  L.curv = mixed ? take(nv) : 0;
  const int u0 = o;  // ---- union: shooting records | QP workspace | RGP workspace
  L.sub = u0;
  const int sub_end = u0 + al4(N * SUBS);
  if (L.gk) { L.r0 = take(nv); L.lb = take(nv); L.ub = take(nv); }
  L.z = take(nv); L.sl = take(nv); L.su = take(nv); L.ll = take(nv); L.lu = take(nv);
  L.dza = take(nv); L.dz = take(nv); L.rho = take(nv); L.act = take(nv); L.rt = take(nv);
  L.grad = take(N * VS);   // slots 10..13 of each stage: d objective / d z
  L.vin = take(N * VS);    // per-sweep input vector: slots 0..3 feed-forward k_i, slots 10..13 sweep-specific
  L.dx = take((N + 1) * VS);
  L.Dx = take((N + 1) * VS);
  // mixed precision: the state trajectory in double, (N + 1) x 16 doubles = exactly the two float vectors dx | Dx, which only the
  // interior point and the float factorisation's gap operand use -- never while dxd is live (polish_mixed).  Index in doubles from the LDS base.
  if (mixed) L.dxd = (L.dbytes + L.dx * 4) / 8;
  if (L.gk) { L.K = gtake(N * KS); L.Linv = gtake(N * 16); }
  else { L.K = take(N * KS); L.Linv = take(N * 16); }
  L.sF = take(4 * VS);
  L.sT = take(4 * VS);   // rows 10..12 of T1'' (+ one row the fp64 hand-over writes unconditionally and nobody reads)
  L.stv = take(2 * VS);
  const int qp_end = o;
  L.rgp = u0;
  const int rgp_end = u0 + al4(3 * nb * nb) + 5 * al4(3 * nb) + 32;
  o = sub_end > qp_end ? sub_end : qp_end;
  if (rgp_end > o) o = rgp_end;
  L.qtotal = o;
  L.gtotal = (g + 15) & ~15;
  return L;
}
template <typename TQ> __host__ __device__ inline size_t lds_bytes(const Lds& L) { return (size_t)CK_HDR + (size_t)L.dbytes + (size_t)L.qtotal * sizeof(TQ); }

// ------------------------------------------------------------------ small helpers

// Ablation build only (BASELINE configs[4] "fp32 vs bf16 tolerance", csrc/Makefile `variant NAME=bf16`): what is STORED
// -- the stage records AB'', gaps, cost gradients, and the RGP mean / covariance between steps -- is rounded to bfloat16
// (8 significant bits, round to nearest even) while the arithmetic stays fp32.  The product never defines MPCQ_BF16_RECORDS.
#ifdef MPCQ_BF16_RECORDS
__device__ inline float st16(float v) {
  unsigned u = (unsigned)__float_as_int(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return __int_as_float((int)(u & 0xFFFF0000u));
}
__device__ inline double st16(double v) { return v; }   // the fp64 path is not part of that ablation
#else
template <typename T> __device__ inline T st16(T v) { return v; }
#endif
template <typename T> __device__ inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ inline T tmax(T a, T b) { return a > b ? a : b; }

// Lane index as an opaque value: everything derived from it (tile coordinates, masks, operand offsets) is then
// recomputed where it is used instead of being hoisted to the top of the kernel and kept in registers through all
// phases (and through every period of the free-running variant).
__device__ inline int lane_id() {
  int t = threadIdx.x;
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(t));
#endif
  return t;
}
// A zero the compiler has to take for a per-lane value.  Added to the index of an LDS broadcast read it keeps everything computed from
// the value on the vector ALU: a wave-uniform value that feeds a select is otherwise moved to SGPRs, at two v_readfirstlane and an
// s_cselect pair per operand (and, scalar floating point not existing on gfx950, comes back through v_mov for every comparison).
__device__ inline int lane_zero() {
  int z = 0;
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(z));
#endif
  return z;
}
// lane broadcast: `lane` must be wave-uniform (a constant after unrolling) -> v_readlane_b32 into an SGPR
__device__ inline int bc(int v, int lane) { CK_EXEC_FULL(1); return __builtin_amdgcn_readlane(v, lane); }
__device__ inline float bc(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline double bc(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// DPP lane permutes inside 16-lane rows (quad_perm / row_ror), no LDS involved
// (old operand undefined + bound_ctrl: every use runs with a full EXEC mask and permutes inside a row, so no source lane is ever
// invalid -- and the compiler needs no copy of the source in front of each v_mov_dpp)
template <int CTRL> __device__ inline int dpp(int v) { CK_EXEC_FULL(2); return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ inline float dpp(float v) { return __int_as_float(dpp<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ inline double dpp(double v) {   // one v_mov_b64_dpp for row_newbcast (the only 64-bit DPP of gfx950), two 32-bit ones otherwise
#if defined(__AMDGCN__) || defined(MPCQ_EMU_BUILD)
  CK_EXEC_FULL(2);
  const long long x = __builtin_bit_cast(long long, v);
  return __builtin_bit_cast(double, (long long)__builtin_amdgcn_update_dpp(0ll, x, CTRL, 0xF, 0xF, true));
#else   // host pass of hipcc (parses device code, never runs it): the builtin is only declared for 32 bits there
  return __hiloint2double(dpp<CTRL>(__double2hiint(v)), dpp<CTRL>(__double2loint(v)));
#endif
}
// DPP permutation taken by the 16-lane rows selected by ROWS (bit r = row r of the wave) only: permute + select.  (The one-instruction
// form -- DPP row_mask with the old value tied to the source -- was 0.6 % faster and was dropped: with it the free-running
// instance of shape (20, 20) stopped reproducing the lockstep launches bit for bit, tests/test_gpu_parity.py::test_config2_full_size,
// although every such instruction was emitted in place)
#ifndef MPCQ_DPP_ROWMASK
template <int CTRL, int ROWS, typename T> __device__ inline T dpp_rows(T v, int h) {   // h = lane >> 4
  const T r = dpp<CTRL>(v);
  return ((ROWS >> h) & 1) ? r : v;
}
#else   // reproducer builds only (tools/repro_codegen): the one-instruction form that was dropped in round 3
template <int CTRL, int ROWS> __device__ inline int dpp_rows(int v, int) { CK_EXEC_FULL(2); return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xF, false); }
template <int CTRL, int ROWS> __device__ inline float dpp_rows(float v, int h) { return __int_as_float(dpp_rows<CTRL, ROWS>(__float_as_int(v), h)); }
template <int CTRL, int ROWS> __device__ inline double dpp_rows(double v, int h) {
  return __hiloint2double(dpp_rows<CTRL, ROWS>(__double2hiint(v), h), dpp_rows<CTRL, ROWS>(__double2loint(v), h));
}
#endif
// sum over the four 16-lane rows (lanes c, c+16, c+32, c+48), result on every lane: after v_permlane16_swap(v, v) the two
// outputs hold rows (0,0,2,2) and (1,1,3,3), after v_permlane32_swap rows (0,1,0,1) and (2,3,2,3) -- their sum is the
// butterfly step on every lane, no select (gfx950; no LDS, no SGPR round trip)
__device__ inline float hsum(float v) {
  CK_EXEC_FULL(3);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned b = (unsigned)__float_as_int(v);
  const u2 r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  const float w = __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
  const unsigned b2 = (unsigned)__float_as_int(w);
  const u2 q = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
  return __int_as_float((int)q[0]) + __int_as_float((int)q[1]);
}
__device__ inline double hsum(double v) {
  CK_EXEC_FULL(3);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const u2 rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double w = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  const unsigned lo2 = (unsigned)__double2loint(w), hi2 = (unsigned)__double2hiint(w);
  const u2 ql = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false), qh = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
  return __hiloint2double((int)qh[0], (int)ql[0]) + __hiloint2double((int)qh[1], (int)ql[1]);
}
// sum over the 16 lanes of a row, result on every lane of the row (DPP only)
template <typename T> __device__ inline T rowsum(T v) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x124>(v);   // row_ror:4
  v += dpp<0x128>(v);   // row_ror:8
  return v;
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

// internal state order of the QP: [q v r p]; o2i(original index) / i2o(internal index)
__host__ __device__ inline int o2i(int o) { return o < 3 ? o + 10 : o - 3; }
__host__ __device__ inline int i2o(int k) { return k < 10 ? k + 3 : k - 10; }

// ---- matrix-core tiles.  One v_mfma_*_16x16x4 per call: D = A(16x4) B(4x16) + C with lane (h = lane>>4,
// c = lane&15) holding A[c][h], B[h][c] and four C/D elements D[RI(reg,h)][c].  A K=16 product is four
// calls; since the k index is only summed over, step s of lane group h may carry ANY k as long as A and
// B agree: we use k = RI(s,h), the same map as the C/D rows, so an accumulator tile can be fed straight
// back as the B operand of the next product (and a symmetric one as the A operand) without leaving
// registers.  RI differs between the f32 and f64 instructions.
template <typename TQ> __device__ inline int RI(int s, int h) { return sizeof(TQ) == 4 ? 4 * h + s : h + 4 * s; }
// A vector held lane-indexed (lane (h,c) holds x[c], the same in every row h) -> group-uniform operand form (every lane of row
// h holds x[RI(s,h)], s = 0..3) without LDS: rotate row h left by U*h lanes (two conditional row_ror; U = 1 for the f64 slot
// map h + 4s, 4 for the f32 map 4h + s), then row_newbcast of the lane that now holds the slot.
// (TQ selects the slot map -- the storage type of the matrix operand --, TV is the type of the vector: the mixed-precision sweeps
//  multiply float operands by a double vector)
template <typename TQ, typename TV> __device__ inline void l2g(TV x, int h, TV (&v)[4]) {
  if (sizeof(TQ) == 8) {
    x = dpp_rows<0x12F, 0xA>(x, h);            // rows 1, 3: row_ror:15 -> lane c <- lane c + 1
    x = dpp_rows<0x12E, 0xC>(x, h);         // rows 2, 3: row_ror:14 -> lane c <- lane c + 2
    v[0] = dpp<0x150>(x); v[1] = dpp<0x154>(x); v[2] = dpp<0x158>(x); v[3] = dpp<0x15C>(x);   // row_newbcast:0,4,8,12
  } else {
    x = dpp_rows<0x12C, 0xA>(x, h);         // rows 1, 3: row_ror:12 -> lane c <- lane c + 4
    x = dpp_rows<0x128, 0xC>(x, h);         // rows 2, 3: row_ror:8  -> lane c <- lane c + 8
    v[0] = dpp<0x150>(x); v[1] = dpp<0x151>(x); v[2] = dpp<0x152>(x); v[3] = dpp<0x153>(x);   // row_newbcast:0..3
  }
}
#ifndef MPCQ_NO_MFMA
__device__ inline void mfma(float (&acc)[4], float a, float b) {
  CK_EXEC_FULL(4);
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 cc = {acc[0], acc[1], acc[2], acc[3]};
  cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, cc, 0, 0, 0);
  acc[0] = cc[0]; acc[1] = cc[1]; acc[2] = cc[2]; acc[3] = cc[3];
}
__device__ inline void mfma(double (&acc)[4], double a, double b) {
  CK_EXEC_FULL(4);
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 cc = {acc[0], acc[1], acc[2], acc[3]};
  cc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cc, 0, 0, 0);
#ifdef MPCQ_DGEMM_SETTLE
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(cc));
#endif
  acc[0] = cc[0]; acc[1] = cc[1]; acc[2] = cc[2]; acc[3] = cc[3];
}
#else
// Ablation build (BASELINE configs[4] "MFMA on/off", csrc/Makefile `variant NAME=nomfma`): the same tile product with
// the matrix cores switched off -- operands fetched from the lanes that hold them through the LDS crossbar
// (ds_bpermute), products on the vector ALU.  D[RI(reg,h)][c] += sum_k A[RI(reg,h)][k] B[k][c], A[r][k] on lane 16k + r,
// B[k][c] on lane 16k + c.  Same results up to the summation order inside one instruction.
template <typename T> __device__ inline void mfma(T (&acc)[4], T a, T b) {
  const int lane = lane_id(), h = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T bk = __shfl(b, 16 * k + c);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[reg] = fma(__shfl(a, 16 * k + RI<T>(reg, h)), bk, acc[reg]);
  }
}
#endif
// A state-sized vector rides in column 14 of a tile: lane (h, 14) holds slots RI(s,h), s = 0..3.
template <typename PT> __device__ inline void vl_load(PT base, int h, float (&v)[4]) {   // slots 4h..4h+3: one 128-bit read
  // (double records read by float arithmetic -- the float interior point of the fp64 instances, solve_qp: two 128-bit reads, rounded once)
  const auto t = ld4(base, 4 * h);
  v[0] = (float)t.a; v[1] = (float)t.b; v[2] = (float)t.c; v[3] = (float)t.d;
}
template <typename PT> __device__ inline void vl_store(PT base, int h, const float (&v)[4]) {
  V4<float> t; t.a = v[0]; t.b = v[1]; t.c = v[2]; t.d = v[3];
  st4(base, 4 * h, t);
}
template <typename PT> __device__ inline void vl_load(PT base, int h, double (&v)[4]) {
  const PT q = base + h;
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = q.c(4 * s);
}
template <typename PT> __device__ inline void vl_store(PT base, int h, const double (&v)[4]) {
  const PT q = base + h;
#pragma unroll
  for (int s = 0; s < 4; ++s) q.c(4 * s) = v[s];
}
// Operand addressing that is valid on every lane (padding lanes read a zero block with stride 0), so
// operand loads are unconditional and can be issued ahead of their stage.  `P` below is the base the
// offsets refer to: the LDS workspace S, or the per-instance global stage record A (GAB layouts).
// stage index x per-lane stride (0 on padding lanes): both far below 2^23 -> one full-rate v_mad_i32_i24 with the offset instead of a
// quarter-rate 32-bit multiply
__device__ inline int mul24(int a, int b) {
#if defined(__AMDGCN__)
  return __mul24(a, b);
#else
  return a * b;
#endif
}
// k-major operand: lane (h,c) <- AB''[RI(s,h)][c]
template <typename TQ> struct KMaj {
  int off[4], str[4];
  __device__ inline KMaj(const Lds& L, int N, int h, int c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = RI<TQ>(s, h);
      off[s] = k < NX ? L.AB + k * ABW + c : L.AB + N * ABS + c;
      str[s] = k < NX ? ABS : 0;
    }
  }
  template <typename PT> __device__ inline void load(PT Pb, int i, TQ (&o)[4]) const {
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = Pb[off[s] + mul24(i, str[s])];
  }
};
// row-major operand: lane (h,c) <- M[c][RI(s,h)] for a row-major matrix of `rows` rows at `base` (stage stride
// `sst`); lanes c >= rows read the zero block at `zero`
template <typename TQ> struct RMaj {
  int off, str, hh;
  __device__ inline RMaj(int zero, int base, int sst, int rows, int h, int c) {
    off = c < rows ? base + c * ABW : zero;
    str = c < rows ? sst : 0;
    hh = h;
  }
  template <typename PT> __device__ inline void load(PT Pb, int i, TQ (&o)[4]) const { vl_load(Pb + (off + mul24(i, str)), hh, o); }
};
// Operands are fetched PD stages ahead of their use: one stage hides the LDS latency, the global stage
// records need more.  q[0] is the current stage; shift() retires it.
#ifndef MPCQ_PD_GLOBAL
#define MPCQ_PD_GLOBAL 2
#endif
template <bool GAB> struct Depth { static constexpr int PD = GAB ? MPCQ_PD_GLOBAL : 1; };
template <typename TQ, int PD> __device__ inline void shift(TQ (&q)[PD + 1][4]) {
#pragma unroll
  for (int d = 0; d < PD; ++d)
#pragma unroll
    for (int s = 0; s < 4; ++s) q[d][s] = q[d + 1][s];
}
__device__ inline float  texp(float x)  { return __expf(x); }
__device__ inline double texp(double x) { return exp(x); }
__device__ inline float  trsqrt(float x)  { return rsqrtf(x); }
__device__ inline double trsqrt(double x) { return 1.0 / sqrt(x); }
// reciprocal of a positive, normal-range pivot: hardware estimate + Newton steps (no scaling / fix-up path)
__device__ inline float  trcp(float x)  { return __fdividef(1.0f, x); }
__device__ inline double trcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);   // v_rcp_f64 is good to ~2^-23: two steps reach the last ulp
  return fma(fma(-x, r, 1.0), r, r);
}
// pivot reciprocal of the stage Hessian: ONE Newton step on v_rcp_f64 (2^-23 -> ~2^-46 = 1.4e-14 relative).  The four pivots of
// a stage are a serial chain (each step two dependent fp64 FMAs at ~16 cycles for a wave on its own); the gains inherit a
// relative error of 1e-14, three orders below the rounding the recursion accumulates over a horizon anyway.
__device__ inline float  trcp1(float x)  { return __fdividef(1.0f, x); }
__device__ inline double trcp1(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ inline float  tabs(float x)  { return fabsf(x); }
__device__ inline double tabs(double x) { return fabs(x); }

template <typename T> __device__ inline void rotmat(const T* q, T* R) {
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz);     R[2] = 2 * (qx * qz + qw * qy);
  R[3] = 2 * (qx * qy + qw * qz);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
  R[6] = 2 * (qx * qz - qw * qy);     R[7] = 2 * (qy * qz + qw * qx);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

// quad constants in the arithmetic type of the caller
template <typename T> struct QC {
  T mass, J[3], tmax, xf[4], yf[4], zl[4], g, imass, iJ[3];
  template <typename M> __device__ inline explicit QC(const M& m) {
    mass = (T)m.mass; tmax = (T)m.tmax; g = (T)m.g; imass = (T)m.imass;
#pragma unroll
    for (int i = 0; i < 3; ++i) { J[i] = (T)m.J[i]; iJ[i] = (T)m.iJ[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xf[i] = (T)m.xf[i]; yf[i] = (T)m.yf[i]; zl[i] = (T)m.zl[i]; }
  }
};

// f(x,u) of the OCP model (src/quad_opt.py:186-251 in the reference); when `sub` != nullptr also
// writes the record the sensitivity pass needs: q(4) r(3) | d vdot/dq (3x4) | d vdot/dv (3x3) | R[:,2]  (SUB_*).
// GP term: m_d(s) = sum_j alpha_dj sf2 exp(-(s - X_j)^2 L2inv / 2), alpha = Kx^-1 mu.
// gd >= 0: this lane sums only GP axis gd (lanes l - gd .. l - gd + 2 hold the three axes of a stage) and the three exchange their sums.
template <typename T, typename TG, typename PA, typename PB, typename PS, typename PG = TG*>
__device__ inline void model_eval(const QC<T>& m, int nb, const TG* L2inv, const TG* sf2, const T* x, const T* u,
                                  PA alpha, PB basis, T* f, PS sub, int gd = -1, PG gx = nullptr) {
  const T* q = x + 3; const T* v = x + 7; const T* r = x + 10;
  T R[9];
  rotmat(q, R);
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
  f[3] = T(0.5) * (-r[0] * q[1] - r[1] * q[2] - r[2] * q[3]);
  f[4] = T(0.5) * (r[0] * q[0] + r[2] * q[2] - r[1] * q[3]);
  f[5] = T(0.5) * (r[1] * q[0] - r[2] * q[1] + r[0] * q[3]);
  f[6] = T(0.5) * (r[2] * q[0] + r[1] * q[1] - r[0] * q[2]);
  const T aT = m.tmax * (u[0] + u[1] + u[2] + u[3]) * m.imass;
  f[7] = R[2] * aT; f[8] = R[5] * aT; f[9] = R[8] * aT - m.g;
  T ty = 0, tx = 0, tz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ty += u[j] * m.yf[j]; tx += u[j] * m.xf[j]; tz += u[j] * m.zl[j]; }
  f[10] = (m.tmax * ty + (m.J[1] - m.J[2]) * r[1] * r[2]) * m.iJ[0];
  f[11] = (-m.tmax * tx + (m.J[2] - m.J[0]) * r[2] * r[0]) * m.iJ[1];
  f[12] = (m.tmax * tz + (m.J[0] - m.J[1]) * r[0] * r[1]) * m.iJ[2];
  T mg[3] = {0, 0, 0}, mp[3] = {0, 0, 0};
  const bool gp = (bool)alpha;
  if (gp) {
    T vb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
    if (gd < 0) {
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
    } else {
      TG s0 = 0, s1 = 0;
      const TG vbd = gd == 0 ? (TG)vb[0] : (gd == 1 ? (TG)vb[1] : (TG)vb[2]);
      const TG l2 = L2inv[gd], sf = sf2[gd];
      for (int j = 0; j < nb; ++j) {
        const TG dlt = vbd - basis[gd * nb + j];
        const TG k = alpha[gd * nb + j] * sf * texp(TG(-0.5) * dlt * dlt * l2);
        s0 += k;
        s1 -= k * dlt;
      }
#if MPCQ_G_SHFL
      // the three lanes of a stage exchange their sums through the LDS crossbar (ds_bpermute: no LDS memory, no barrier; lane 63, which has
      // no partners, reads lanes 0 / 1 and is not used)
      const int l0 = lane_id() - gd;
      const TG e0 = s0, e1 = s1 * l2;
      CK_EXEC_FULL(5);
#pragma unroll
      for (int d = 0; d < 3; ++d) { mg[d] = (T)__shfl(e0, l0 + d); mp[d] = (T)__shfl(e1, l0 + d); }
#else
      gx[gd] = s0;
      gx[3 + gd] = s1 * l2;
      __syncthreads();
#pragma unroll
      for (int d = 0; d < 3; ++d) { mg[d] = (T)gx[d]; mp[d] = (T)gx[3 + d]; }
      __syncthreads();
#endif
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) f[7 + i] += R[3 * i] * mg[0] + R[3 * i + 1] * mg[1] + R[3 * i + 2] * mg[2];
  }
  if (!sub) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) sub[SUB_Q + i] = q[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) sub[SUB_R + i] = r[i];
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
      sub[SUB_JQ + row * 4 + i] = 2 * val;
    }
  }
#pragma unroll
  for (int row = 0; row < 3; ++row)
#pragma unroll
    for (int col = 0; col < 3; ++col)
      sub[SUB_JV + row * 3 + col] = gp ? (R[3 * row] * mp[0] * R[3 * col] + R[3 * row + 1] * mp[1] * R[3 * col + 1] +
                                      R[3 * row + 2] * mp[2] * R[3 * col + 2])
                                   : T(0);
  sub[SUB_RZ] = R[2]; sub[SUB_RZ + 1] = R[5]; sub[SUB_RZ + 2] = R[8];
}

// one RK4 step of the NOMINAL model in double (quad_optimizer.discrete_dynamics on quad_nominal)
template <typename M>
MPCQ_PHASE void rk4_nominal(const M& m, const double* x, const double* u, double dt, double* xo) {
  const QC<double> qc(m);
  double k[NX], xt[NX], acc[NX];
  const double* nul = nullptr;
#pragma unroll
  for (int i = 0; i < NX; ++i) { xt[i] = x[i]; acc[i] = 0; }
  MPCQ_RK_LOOP
  for (int s = 0; s < 4; ++s) {   // k1..k4: acc = k1 + 2 k2 + 2 k3 + k4, next point x + {dt/2, dt/2, dt} k
    model_eval<double, double>(qc, 0, nul, nul, xt, u, nul, nul, k, (double*)nullptr);
    const double wa = (s == 0 || s == 3) ? 1.0 : 2.0, hc = s == 2 ? dt : dt / 2;
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += wa * k[i]; xt[i] = x[i] + hc * k[i]; }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = x[i] + dt / 6 * acc[i];
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
  const double aT = m.tmax * (u[0] + u[1] + u[2] + u[3]) * m.imass;
  double vb[3], ad[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sg = (double)((vb[i] > 0) - (vb[i] < 0));
    ad[i] = (-m.aero_drag * vb[i] * vb[i] * sg - m.rotor_drag[i] * vb[i]) * m.imass;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) f[7 + i] = R[3 * i] * ad[0] + R[3 * i + 1] * ad[1] + R[3 * i + 2] * (ad[2] + aT);
  f[9] -= m.g;
  double ty = 0, tx = 0, tz = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ty += u[j] * m.yf[j]; tx += u[j] * m.xf[j]; tz += u[j] * m.zl[j]; }
  f[10] = (m.tmax * ty + (m.J[1] - m.J[2]) * r[1] * r[2]) * m.iJ[0];
  f[11] = (-m.tmax * tx + (m.J[2] - m.J[0]) * r[2] * r[0]) * m.iJ[1];
  f[12] = (m.tmax * tz + (m.J[0] - m.J[1]) * r[0] * r[1]) * m.iJ[2];
}
template <typename M>
MPCQ_PHASE void plant_rk4(const M& m, double* x, const double* uin, double dt) {
  double u[4], k[NX], xt[NX], acc[NX];
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = tmin(1.0, tmax(0.0, uin[j]));
#pragma unroll
  for (int i = 0; i < NX; ++i) { xt[i] = x[i]; acc[i] = 0; }
  MPCQ_RK_LOOP
  for (int s = 0; s < 4; ++s) {
    plant_eval(m, xt, u, k);
    const double wa = (s == 0 || s == 3) ? 1.0 : 2.0, hc = s == 2 ? dt : dt / 2;
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += wa * k[i]; xt[i] = x[i] + hc * k[i]; }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = x[i] + dt / 6 * acc[i];
}

// ------------------------------------------------------------------ kernel configurations
// One step-kernel instantiation per Cfg.  N = 0 / NB = -1 read the horizon and the RGP basis size from the
// model at run time (any shape); fixed values turn every LDS offset, trip count and index division into a
// compile-time constant for the shapes that matter (see mpcq_api.hip for the table of instances).
template <typename T_, bool GAB_, int N_ = 0, int NB_ = -1, bool RUN_ = false, bool GK_ = false> struct Cfg {
  using T = T_;
  static constexpr bool GAB = GAB_;
  static constexpr bool GK = GK_;     // compact layout (lds_layout gab = 2): gains in the global record, r0 / lb / ub written behind the shooting
  static_assert(!GK_ || GAB_, "the compact layout keeps the stage records in global memory as well");
  static constexpr int LAYOUT = GK_ ? 2 : (GAB_ ? 1 : 0);
  static constexpr bool RUN = RUN_;   // free-running closed loop: the kernel iterates over control periods
  static constexpr int N = N_, NB = NB_;
};
template <typename C, typename M> __device__ inline int cN(const M& m) { return C::N > 0 ? C::N : m.N; }
template <typename C, typename M> __device__ inline int cNB(const M& m) { return C::NB >= 0 ? C::NB : m.nb; }

// ------------------------------------------------------------------ shooting
// pass 1: lane (triple) per interval, 4 RK substages in TQ; writes records + gap c_i = Phi_i - X_{i+1}
// (the part X_i - X_{i+1} of the gap is formed in double)
// TS: arithmetic of the integration.  The float instances integrate in double too (MPCQ_MIXED_SHOOT64) and round the RECORDS to
// float: their QP solution is refined against fp64 residuals of the stored stage data, so what is left of the 1e-4 budget goes to
// the data themselves -- records computed in float carry ~1e-6 relative error, which a saturated quadrotor far off its reference
// (gradient scale 3e4) turns into 2e-4 .. 7e-4 of control deviation; computed in double and rounded once it is the 6e-8 of the storage.
// The GP sums stay in TQ (the exps of the basis): benign in float.
#ifndef MPCQ_MIXED_SHOOT64
#define MPCQ_MIXED_SHOOT64 1
#endif
template <typename TQ> struct ShootT { using T = TQ; };
#if MPCQ_MIXED_SHOOT64
template <> struct ShootT<float> { using T = double; };
#endif
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_PHASE void shoot_states(const DevModel<TQ>& m, P<double> D, P<TQ> S, P<TQ> A, const Lds& L, bool gp) {
  using TS = typename ShootT<TQ>::T;
  const int N = cN<C>(m), lane = lane_id();
  const QC<TS> qc(m);
  const TS h = (TS)m.h;
  // with the GP in the model the three axis sums (nb exps each) of a stage go to three neighbouring lanes
  const int per = gp ? 3 : 1, lanes_used = gp ? 63 : 64, spr = lanes_used / per;   // stages per round
  P<TQ> gx = S + (L.gx + (lane / 3) * 8);   // LDS exchange scratch (MPCQ_G_SHFL = 0 only)
  for (int base = 0; base < N; base += spr) {
    const int il = lane / per, d = lane - il * per;
    const bool valid = lane < lanes_used && base + il < N;
    const int i = valid ? base + il : 0;
    const int gd = gp ? (lane < lanes_used ? d : 0) : -1;
    TS x[NX], u[NU], k[NX], xt[NX], acc[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = (TS)D[L.X + i * NX + j];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = (TS)D[L.U + i * NU + j];
    const P<TQ> al = gp ? S + L.alpha : P<TQ>(nullptr);
    const P<TQ> sub = (valid && d == 0) ? S + (L.sub + i * SUBS) : P<TQ>(nullptr);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = 0; xt[j] = x[j]; }
    MPCQ_RK_LOOP
    for (int s = 0; s < 4; ++s) {   // acc = k1 + 2 k2 + 2 k3 + k4, next point x + {h/2, h/2, h} k
      model_eval<TS, TQ>(qc, cNB<C>(m), m.L2inv, m.sf2, xt, u, al, S + L.basis, k, sub ? sub + s * SUBW : P<TQ>(nullptr), gd, gx);
      const TS wa = (s == 0 || s == 3) ? TS(1) : TS(2), hc = s == 2 ? h : h / 2;
#pragma unroll
      for (int j = 0; j < NX; ++j) { acc[j] += wa * k[j]; xt[j] = x[j] + hc * k[j]; }
    }
    if (valid && d == 0) {
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const double gap = (D[L.X + i * NX + j] - D[L.X + (i + 1) * NX + j]) + (double)(h / 6 * acc[j]);
        A[L.c + i * VS + o2i(j)] = st16((TQ)gap);
      }
    }
  }
}
// pass 2: item = (interval i, column j of [A|B], j = 3..16) -> AB'[i][r][j-3]; arithmetic TQ, records TR
template <typename C, typename TQ, typename TR, typename M>
__device__ inline void shoot_sens_t(const M& m, P<TR> S, P<TR> A, const Lds& L, const int N) {
  const QC<TQ> qc(m);
  const TQ h = (TQ)m.h;
  const TQ c10 = (qc.J[1] - qc.J[2]) * qc.iJ[0], c11 = (qc.J[2] - qc.J[0]) * qc.iJ[1], c12 = (qc.J[0] - qc.J[1]) * qc.iJ[2];
  const TQ tm = qc.tmax * qc.imass;
  for (int it = lane_id(); it < N * 14; it += 64) {
    const int i = it / 14, jp = it - i * 14, j = 3 + jp;
    const bool ucol = j >= NX;
    TQ Sp[NX], acc[NX], Z[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) { Sp[r] = 0; acc[r] = 0; }
    TQ jur[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < NU; ++c)
      if (j - NX == c) { jur[0] = qc.tmax * qc.yf[c] * qc.iJ[0]; jur[1] = -qc.tmax * qc.xf[c] * qc.iJ[1]; jur[2] = qc.tmax * qc.zl[c] * qc.iJ[2]; }
    MPCQ_SENS_LOOP
    for (int s = 0; s < 4; ++s) {
      const P<TR> sub = S + (L.sub + i * SUBS + s * SUBW);
      const TQ hs = s == 0 ? TQ(0) : (s == 3 ? h : h * TQ(0.5)), ws = (s == 0 || s == 3) ? TQ(1) : TQ(2);
#pragma unroll
      for (int r = 0; r < NX; ++r) Z[r] = ((r == j) ? TQ(1) : TQ(0)) + hs * Sp[r];
      const TQ qw = (TQ)sub[SUB_Q], qx = (TQ)sub[SUB_Q + 1], qy = (TQ)sub[SUB_Q + 2], qz = (TQ)sub[SUB_Q + 3];
      const TQ r0 = (TQ)sub[SUB_R], r1 = (TQ)sub[SUB_R + 1], r2 = (TQ)sub[SUB_R + 2];
      TQ Sn[NX];
      Sn[0] = Z[7]; Sn[1] = Z[8]; Sn[2] = Z[9];
      Sn[3] = TQ(0.5) * (-r0 * Z[4] - r1 * Z[5] - r2 * Z[6] - qx * Z[10] - qy * Z[11] - qz * Z[12]);
      Sn[4] = TQ(0.5) * (r0 * Z[3] + r2 * Z[5] - r1 * Z[6] + qw * Z[10] - qz * Z[11] + qy * Z[12]);
      Sn[5] = TQ(0.5) * (r1 * Z[3] - r2 * Z[4] + r0 * Z[6] + qz * Z[10] + qw * Z[11] - qx * Z[12]);
      Sn[6] = TQ(0.5) * (r2 * Z[3] + r1 * Z[4] - r0 * Z[5] - qy * Z[10] + qx * Z[11] + qw * Z[12]);
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        TQ t = ucol ? (TQ)sub[SUB_RZ + row] * tm : TQ(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) t += (TQ)sub[SUB_JQ + row * 4 + c] * Z[3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) t += (TQ)sub[SUB_JV + row * 3 + c] * Z[7 + c];
        Sn[7 + row] = t;
      }
      Sn[10] = jur[0] + c10 * (r2 * Z[11] + r1 * Z[12]);
      Sn[11] = jur[1] + c11 * (r2 * Z[10] + r0 * Z[12]);
      Sn[12] = jur[2] + c12 * (r1 * Z[10] + r0 * Z[11]);
#pragma unroll
      for (int r = 0; r < NX; ++r) { acc[r] += ws * Sn[r]; Sp[r] = Sn[r]; }
    }
    const P<TR> AB = A + (L.AB + i * ABS);
#pragma unroll
    for (int r = 0; r < NX; ++r) AB[o2i(r) * ABW + jp] = st16((TR)(((r == j) ? TQ(1) : TQ(0)) + h / 6 * acc[r]));
  }
  // zero the two pad columns (read by the vectorised 4-wide loads)
  for (int it = lane_id(); it < N * NX * 2; it += 64) A[L.AB + (it >> 1) * ABW + 14 + (it & 1)] = 0;
  for (int it = lane_id(); it < N * 3; it += 64) A[L.c + (it / 3) * VS + NX + it % 3] = 0;
}

template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_PHASE void shoot_sens(const DevModel<TQ>& m, P<TQ> S, P<TQ> A, const Lds& L) {
  // The float instances run this pass in double as well (records rounded to float once).  What the double integration buys first is the
  // gap c_i of shoot_states (the residual of the dynamics, which the QP solution follows one to one: 7e-4 -> 3e-7 on a saturated quadrotor);
  // the sensitivities shape the QP's curvature, and computed in float they cost another factor 3-5 where the horizon is long and the inputs
  // idle near zero (N = 50, bench workload: worst deviation 9.5e-6 against 2.0e-6 of full thrust).  What is left is the float STORAGE of
  // AB'' (6e-8 relative): keeping what the float gaps and cost gradients drop of their double values as a second float, read by the
  // double sweeps only, was measured and changed nothing (7.8e-5 against 9.2e-5 of an idling quadrotor's own largest control) at 5-10 % of
  // the speed.  -DMPCQ_MIXED_SENS32: this pass in float (2-3 % faster).
#ifdef MPCQ_MIXED_SENS32
  shoot_sens_t<C, TQ, TQ>(m, S, A, L, cN<C>(m));
#else
  shoot_sens_t<C, typename ShootT<TQ>::T, TQ>(m, S, A, L, cN<C>(m));
#endif
}

// ------------------------------------------------------------------ QP: vector sweeps
// QP in (dx, du): min sum_i 1/2 dx'Q_i dx + qv_i'dx + 1/2 du'R du + r0_i'du  s.t. dx_{i+1} = A dx_i + B du_i + c_i,
// dx_0 = x0 - X_0, lb <= du <= ub.  Q_i = h W_x (i<N) / W_e, R = h W_u diagonal; qv, r0, lb, ub, c prepared by
// the caller (internal state order).  With position last, [A|B] = [AB''(:,0:10) | [0;I] | AB''(:,10:14)]: the 14
// columns of AB'' (10 states q,v,r + 4 inputs) fit one 16-wide matrix-core tile and the position columns
// are handled as identity.  The factorisation uses tile products; the sweeps multiply the same operand registers by ONE vector on
// the vector ALU (riccati_forward).

// per-lane slot classes of the four registers of a column-14 vector, as 0/1 multipliers
template <typename TQ> struct Sel {
  TQ A[4], P[4], U[4], K[4];   // slot < 10 (q,v,r) | position 10..12 | input slots 10..13 | slots 0..3
  __device__ inline Sel(int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = RI<TQ>(s, h);
      A[s] = slot < 10 ? TQ(1) : TQ(0);
      P[s] = (slot >= 10 && slot < NX) ? TQ(1) : TQ(0);
      U[s] = (slot >= 10 && slot < 14) ? TQ(1) : TQ(0);
      K[s] = slot < NU ? TQ(1) : TQ(0);
    }
  }
};
// compile-time position of QP-input j inside a column-14 vector: slot 10 + j = RI(s, h)
template <typename TQ> __device__ constexpr int in_h(int j) { return sizeof(TQ) == 4 ? (10 + j) / 4 : (10 + j) % 4; }
template <typename TQ> __device__ constexpr int in_s(int j) { return sizeof(TQ) == 4 ? (10 + j) % 4 : (10 + j) / 4; }
__host__ __device__ inline int GI(int i) { return (i >> 2) * VS + 10 + (i & 3); }   // input i of the nv-vector inside a 16-stride array

// forward rollout dx_{i+1} = A dx_i + B z_i (+ c_i); dx_0 taken from S[dxo + 0..15]
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_COLD void rollout(const DevModel<TQ>& m, P<TQ> S, P<TQ> A, const Lds& L, int dxo, int zo, bool with_c) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15, nv = N * NU;
  for (int i = lane; i < nv; i += 64) S[L.vin + GI(i)] = S[zo + i];
  __syncthreads();
  {   // vector-ALU form (see riccati_forward)
    const Sel<TQ> sel(h);
    const RMaj<TQ> rm(L.AB + N * ABS, L.AB, ABS, NX, h, c);
    constexpr int PD = Depth<GAB>::PD;
    const bool prow = c >= 10 && c < NX;
    TQ xc = S[dxo + c], qa[PD + 1][4], qz[PD + 1][4], qc[PD + 1];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int id = d < N ? d : N - 1;
      rm.load(A, id, qa[d]);
      vl_load(S + L.vin + id * VS, h, qz[d]);
      qc[d] = A[L.c + id * VS + c];
    }
#pragma unroll MPCQ_UNROLL_SWEEP
    for (int i = 0; i < N; ++i) {
      const int ip = i + PD < N ? i + PD : N - 1;
      rm.load(A, ip, qa[PD]);
      vl_load(S + L.vin + ip * VS, h, qz[PD]);
      qc[PD] = A[L.c + ip * VS + c];
      TQ xv[4], vB[4];
      l2g<TQ>(xc, h, xv);
#pragma unroll
      for (int s = 0; s < 4; ++s) vB[s] = sel.A[s] * xv[s] + sel.U[s] * qz[0][s];
      const TQ t = hsum((qa[0][0] * vB[0] + qa[0][1] * vB[1]) + (qa[0][2] * vB[2] + qa[0][3] * vB[3]));
      TQ xn = t + (prow ? xc : TQ(0)) + (with_c ? qc[0] : TQ(0));
      xn = c < NX ? xn : TQ(0);
      xc = xn;
      if (lane < VS) S[dxo + (i + 1) * VS + lane] = xn;
      shift<TQ, PD>(qa); shift<TQ, PD>(qz);
#pragma unroll
      for (int d = 0; d < PD; ++d) qc[d] = qc[d + 1];
    }
    __syncthreads();
  }
}

// adjoint sweep: grad = d/dz of the QP objective at (dx(z), z)
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_COLD void adjoint(const DevModel<TQ>& m, P<TQ> S, P<TQ> A, const Lds& L) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15, nv = N * NU;
  for (int i = lane; i < nv; i += 64) S[L.vin + GI(i)] = S[L.wq + 2 * VS + (i & 3)] * S[L.z + i] + S[L.r0 + i];
  __syncthreads();
  {   // vector-ALU form (see riccati_forward)
    const KMaj<TQ> km(L, N, h, c);
    constexpr int PD = Depth<GAB>::PD;
    const bool arow = c < 10, prow = c >= 10 && c < NX;
    const TQ qdc = S[L.wq + c];
    TQ pc = S[L.wq + VS + c] * S[L.dx + N * VS + c] + A[L.qv + N * VS + c];
    TQ qa[PD + 1][4], qq[PD + 1];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int id = N - 1 - d > 0 ? N - 1 - d : 0;
      km.load(A, id, qa[d]);
      qq[d] = A[L.qv + id * VS + c];
    }
#pragma unroll MPCQ_UNROLL_SWEEP
    for (int i = N - 1; i >= 0; --i) {
      const int ip = i - PD > 0 ? i - PD : 0;
      km.load(A, ip, qa[PD]);
      qq[PD] = A[L.qv + ip * VS + c];
      const TQ dxc = S[L.dx + i * VS + c], gvc = S[L.vin + i * VS + c];
      TQ pi[4];
      l2g<TQ>(pc, h, pi);
      const TQ t = hsum((qa[0][0] * pi[0] + qa[0][1] * pi[1]) + (qa[0][2] * pi[2] + qa[0][3] * pi[3]));   // (AB''^T pi)[c]
      if (lane < VS) S[L.grad + i * VS + lane] = t + gvc;
      pc = (arow ? t : (prow ? pc : TQ(0))) + (qdc * dxc + qq[0]);
      shift<TQ, PD>(qa);
#pragma unroll
      for (int d = 0; d < PD; ++d) qq[d] = qq[d + 1];
    }
    __syncthreads();
  }
}

// backward vector recursion with stored K, Linv: feed-forward k_i (into S[L.vin] slots 0..3) for linear term rho
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB, typename M = DevModel<TQ>, typename PA = P<TQ>>
MPCQ_COLD void riccati_backward_vec(const M& m, P<TQ> S, PA A, P<TQ> Kb, const Lds& L, bool polish) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15, nv = N * NU;
  for (int i = lane; i < nv; i += 64) S[L.vin + GI(i)] = S[L.rho + i];
  __syncthreads();
  {   // vector-ALU form (see riccati_forward)
    const KMaj<TQ> km(L, N, h, c);
    const int lj = lane < NU ? lane : 0;
    constexpr int PD = Depth<GAB>::PD;
    const bool arow = c < 10, prow = c >= 10 && c < NX;
    // gains: from LDS at their stage, from the global record (compact layout) as far ahead as the stage records
    constexpr int KD = C::GK ? PD : 0;
    TQ pc = 0, qa[PD + 1][4], kq[KD + 1];
    V4<TQ> lq[KD + 1];
#pragma unroll
    for (int d = 0; d < PD; ++d) km.load(A, N - 1 - d > 0 ? N - 1 - d : 0, qa[d]);
#pragma unroll
    for (int d = 0; d < KD; ++d) {
      const int id = N - 1 - d > 0 ? N - 1 - d : 0;
      kq[d] = Kb[L.K + id * KS + h * ABW + c];
      lq[d] = ld4(Kb, L.Linv + id * 16 + lj * 4);
    }
#pragma unroll MPCQ_UNROLL_SWEEP
    for (int i = N - 1; i >= 0; --i) {
      km.load(A, i - PD > 0 ? i - PD : 0, qa[PD]);
      {
        const int ik = i - KD > 0 ? i - KD : 0;
        kq[KD] = Kb[L.K + ik * KS + h * ABW + c];              // K[h][c]
        lq[KD] = ld4(Kb, L.Linv + ik * 16 + lj * 4);
      }
      const TQ rvc = S[L.vin + i * VS + c];                    // rho_j on lane column 10 + j
      const TQ kk = kq[0];
      const V4<TQ> li = lq[0];
      const TQ rtj = S[L.rt + i * NU + lj];
      TQ pv[4];
      l2g<TQ>(pc, h, pv);
      const TQ t = hsum((qa[0][0] * pv[0] + qa[0][1] * pv[1]) + (qa[0][2] * pv[2] + qa[0][3] * pv[3]));   // (AB''^T p)[c]
      const TQ gt = t + rvc;                                   // gt_j = rho_j + (B^T p)_j on lane column 10 + j
      // row h needs gt_h on every lane: rotate row h left by h lanes, broadcast lane 10
      TQ r = dpp_rows<0x12F, 0xA>(gt, h);
      r = dpp_rows<0x12E, 0xC>(r, h);
      const TQ gh = dpp<0x15A>(r);
      pc = (arow ? t : (prow ? pc : TQ(0))) + hsum(kk * gh);   // p_i = A^T p_{i+1} + K^T gt (pinned rows of K are 0)
      TQ g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = bc(gt, 10 + j);
      if (lane < NU) {
        const TQ kvj = -(li.a * g[0] + li.b * g[1] + li.c * g[2] + li.d * g[3]);
        S[L.vin + i * VS + lane] = (polish && rtj < TQ(0)) ? TQ(0) : kvj;
      }
      shift<TQ, PD>(qa);
#pragma unroll
      for (int d = 0; d < KD; ++d) { kq[d] = kq[d + 1]; lq[d] = lq[d + 1]; }
    }
    __syncthreads();
  }
}

// forward sweep: Dx_0 = 0; dz_i = K_i Dx_i + k_i ; Dx_{i+1} = A Dx_i + B dz_i   (out: S[dzo], S[L.Dx])
// affine: the sweep of the affine recursion instead: starts from dx_0 in S[L.dx], adds the gaps (S[L.Dx]) and writes the
// state trajectory to S[L.dx] (z_i = K_i dx_i + k_i to S[dzo])
// A matrix-vector product uses one of the 16 columns of a tile, and v_mfma_f64_16x16x4 occupies the matrix pipe for 16 passes
// whatever the columns hold: the sweeps do their two products per stage on the vector ALU instead, with the SAME operand
// registers (lane (h,c) holds M[c][RI(s,h)], s = 0..3): four FMAs against the group-uniform vector slots x[RI(s,h)], then
// the sum over the four lane rows (hsum).  The result arrives lane-indexed (lane (.,c) holds row c); the next product needs
// it group-uniform again: the inputs go through four v_readlane, the state through the LDS vector that the sweep writes anyway.
template <typename C, bool affine = false, typename TQ = typename C::T, bool GAB = C::GAB, typename M = DevModel<TQ>, typename PA = P<TQ>>
MPCQ_PHASE void riccati_forward(const M& m, P<TQ> S, PA A, P<TQ> Kb, const Lds& L, int dzo PF_ARG) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15;
  constexpr bool F64 = sizeof(TQ) == 8;
  const Sel<TQ> sel(h);
  const RMaj<TQ> rm(L.AB + N * ABS, L.AB, ABS, NX, h, c);
  constexpr int PD = Depth<GAB>::PD;
  const int xo = affine ? L.dx : L.Dx;
  const bool prow = c >= 10 && c < NX;   // position rows: identity block of [A|B]
  // Input slots 10 + j of the operand held by lane row h: f64 (slot = h + 4s) one per row -- register 2 (h >= 2) or 3, input
  // (h + 2) & 3; f32 (slot = 4h + s) two in rows 2 and 3 -- registers 2,3 / 0,1, inputs 0,1 / 2,3.  Row h computes just those
  // dz_j = K_j . Dx + k_j itself, from the lane-indexed Dx (K row j lane-indexed, sum over the row): no broadcast of dz at all.
  const int j0 = F64 ? ((h + 2) & 3) : (h == 3 ? 2 : 0), j1 = F64 ? j0 : j0 + 1;
  const bool hasu = F64 || h >= 2;
  const int ko0 = L.K + j0 * ABW + c, ko1 = L.K + j1 * ABW + c;
  // gain rows: one stage ahead from LDS; from the global record (compact layout) as far ahead as the stage records
  constexpr int KD = C::GK ? PD : 1;
  TQ xc = 0, qa[PD + 1][4], kq0[KD + 1], kq1[KD + 1], g0, g1 = 0, gc = 0, gcn = 0;
  if (affine) { xc = S[L.dx + c]; gc = S[L.Dx + c]; }
  else if (lane < VS) S[L.Dx + lane] = 0;
#pragma unroll
  for (int d = 0; d < PD; ++d) rm.load(A, d < N ? d : N - 1, qa[d]);
#pragma unroll
  for (int d = 0; d < KD; ++d) {
    const int id = d < N ? d : N - 1;
    kq0[d] = Kb[ko0 + id * KS];
    kq1[d] = F64 ? TQ(0) : Kb[ko1 + id * KS];
  }
  g0 = S[L.vin + j0];
  if (!F64) g1 = S[L.vin + j1];
#pragma unroll MPCQ_UNROLL_SWEEP
  for (int i = 0; i < N; ++i) {
    const int ip = i + 1 < N ? i + 1 : i, ig = i + PD < N ? i + PD : N - 1, ik = i + KD < N ? i + KD : N - 1;
    rm.load(A, ig, qa[PD]);
    kq0[KD] = Kb[ko0 + ik * KS];
    kq1[KD] = F64 ? TQ(0) : Kb[ko1 + ik * KS];
    const TQ k0 = kq0[0], k1 = kq1[0];
    const TQ g0n = S[L.vin + ip * VS + j0];
    TQ g1n = 0;
    if (!F64) g1n = S[L.vin + ip * VS + j1];
    if (affine) gcn = S[L.Dx + ip * VS + c];
    PF_FINE(11);
    const TQ u0 = rowsum(k0 * xc) + g0;
    TQ u1 = 0;
    if (!F64) u1 = rowsum(k1 * xc) + g1;
    PF_FINE(12);
    TQ xv[4];
    l2g<TQ>(xc, h, xv);
    TQ ta = (qa[0][0] * (sel.A[0] * xv[0]) + qa[0][1] * (sel.A[1] * xv[1])) + (qa[0][2] * (sel.A[2] * xv[2]) + qa[0][3] * (sel.A[3] * xv[3]));
    PF_FINE(13);
    if (F64) {
      ta += (h >= 2 ? qa[0][2] : qa[0][3]) * u0;
      if (c == 0) S[dzo + i * NU + j0] = u0;
    } else {
      if (hasu) ta += (h == 2 ? qa[0][2] : qa[0][0]) * u0 + (h == 2 ? qa[0][3] : qa[0][1]) * u1;
      if (c == 0 && hasu) { S[dzo + i * NU + j0] = u0; S[dzo + i * NU + j1] = u1; }
    }
    TQ xn = hsum(ta) + (prow ? xc : TQ(0)) + (affine ? gc : TQ(0));
    xn = c < NX ? xn : TQ(0);
    PF_FINE(14);
    xc = xn;
    if (lane < VS) S[xo + (i + 1) * VS + lane] = xn;
    g0 = g0n; g1 = g1n; gc = gcn;
    shift<TQ, PD>(qa);
#pragma unroll
    for (int d = 0; d < KD; ++d) { kq0[d] = kq0[d + 1]; kq1[d] = kq1[d + 1]; }
  }
  __syncthreads();
}

// A pinned input is eliminated from a stage by giving it an astronomically large diagonal entry instead of masking its row and
// column: with Lambda_aa = D the LDL^T leaves l_ka = Lambda_ka / D (~1e-300), the other pivots change by Lambda_ka^2 / D (far below
// one ulp), and the gain row K_a = O(1 / D) -- the same numbers as the masked elimination after rounding, without the mask
// arithmetic on the critical path of every stage.  The pinned input itself is held at its bound by the caller.
template <typename TQ> __device__ inline TQ pin_diag() { return sizeof(TQ) == 8 ? TQ(1e300) : TQ(1e30f); }

// ------------------------------------------------------------------ QP: Riccati factorisation
// Backward sweep recomputing P_i, K_i, Lambda_i^-1 for R~ = R + (polish ? 0 : ll/sl + lu/su) with inputs
// pinned by `act` eliminated in polish mode, merged with the vector recursion for the linear term rho
// (the vector rides in pad column 14 of the second product).  Per stage, on the matrix cores:
//   T1'' = P_{i+1} AB''            (P is kept in registers as an accumulator tile and, being symmetric,
//   F''  = AB''^T [T1'' | p]        is fed back as the A operand)
//   P_i  = Q + G + M^T K           with G = [A|B]^T P [A|B] restricted to states, assembled from F'', T1'', P.
// The 4x4 stage Hessian Lambda = R~ + F''[10:14,10:14] is factorised in registers (Cholesky) by the lanes
// that need it.  Returns false if a stage Hessian was not positive definite.
template <typename C, bool polish, bool affine = false, typename TQ = typename C::T, bool GAB = C::GAB, typename M = DevModel<TQ>, typename PA = P<TQ>>
MPCQ_PHASE bool riccati_factor(const M& m, P<TQ> S, PA A, P<TQ> Kb, const Lds& L PF_ARG, TQ* gscale = nullptr, P<TQ> mrows = nullptr,
                                      P<TQ> pstore = nullptr, int start = -1, bool have_rt = false, const int pst_valid = 1 << 30, const int pst_store = 1 << 30) {
  const int N = cN<C>(m), lane = lane_id(), nv = N * NU, h = lane >> 4, c = lane & 15;
  const bool vl = c == 14;
  bool ok = true;
  // stage input Hessian diagonals R~ (negative value = input pinned by the polish); have_rt: the caller has written them
  for (int i = lane; i < nv && !have_rt; i += 64) {
    const TQ rr = S[L.wq + 2 * VS + (i & 3)];
    TQ v;
    if (!polish) v = rr + S[L.ll + i] * trcp(S[L.sl + i]) + S[L.lu + i] * trcp(S[L.su + i]);
    else v = S[L.act + i] != TQ(0) ? -pin_diag<TQ>() : rr;   // sign = pinned flag, magnitude = the diagonal the stage Hessian gets
    S[L.rt + i] = v;
  }
  const Sel<TQ> sel(h);
  const KMaj<TQ> km(L, N, h, c);
  // P_N = W_e as an accumulator tile: Pop[s] = P[RI(s,h)][c]; per-lane masks for the assembly of Q + G
  TQ Pop[4], pv[4] = {0, 0, 0, 0}, qdg[4], mA2[4], mA1[4], mPo[4], mT[4], gmax = 0;
  if (affine) vl_load(A + L.qv + N * VS, h, pv);   // p_N = q_N
  // Restart: the recursion above the highest stage whose working set changed is unchanged, so it resumes from the
  // cost-to-go (P_{start+1}, p_{start+1}) the previous factorisation stored (K_i, Lambda_i^-1, k_i of the stages
  // above are still in LDS).
  // (pst_valid: highest stage whose stored tile is current; pst_store: tiles are kept for stages up to this one only -- the caller knows
  //  above which stage the working set cannot change, and a cost-to-go tile is 2 KB of global stores)
  const int first = pstore ? pst_first(N, start, pst_valid) : N - 1;
  const bool resumed = first < N - 1;
  int toff[4];
  const bool cq = c < 10, cp = c >= 10 && c < NX;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = RI<TQ>(s, h);
    const bool rq = row < 10, rp = row >= 10 && row < NX;
    Pop[s] = (row == c && c < NX) ? S[L.wq + VS + c] : TQ(0);
    if (resumed) { Pop[s] = pstore[(first / PSTEP) * PST + lane * 4 + s]; pv[s] = pstore[(first / PSTEP) * PST + 256 + (row < VS ? row : 0)]; }   // tile of stage first + 1
    qdg[s] = (row == c && c < NX) ? S[L.wq + c] : TQ(0);
    mA2[s] = (cq && rq) ? TQ(1) : TQ(0);
    mA1[s] = (cq && rp) ? TQ(1) : TQ(0);
    mPo[s] = (cp && rp) ? TQ(1) : TQ(0);
    mT[s] = (cp && rq) ? TQ(1) : TQ(0);
    toff[s] = L.sT + (cp ? (c - 10) * VS : 0) + (row < VS ? row : 0);
  }
  // Roles in the solve behind the stage Hessian: EVERY lane row solves column c of M (c < 13; M[j][c] = F''[10+j][c] for c < 10,
  // T1''[c][10+j] for the position columns), so lane (h,c) has both operands of the k=4 tile M^T K -- M[h][c] and K[h][c] -- in
  // registers, and p_i[c] lane-indexed: nothing of the P update waits for an LDS round trip.  Lanes (0,13), (0,14), (0,15), (1,13)
  // solve for the unit vectors: rows 0..3 of Lambda^-1 and the feed-forward k_i.
  const TQ mmask = c < NX ? TQ(1) : TQ(0);
  const int b3 = c < NX ? c : 0;
  const bool inv = c >= NX && (h == 0 || (h == 1 && c == NX));
  const int vj = h == 0 ? (c >= NX ? c - NX : 0) : 3;
  const int m3off = b3 < 10 ? L.sF + b3 : L.sT + (b3 - 10) * VS + 10, m3str = b3 < 10 ? VS : 1;   // M[j][b3]
  const int tboff = b3 < 10 ? L.stv + VS + b3 : L.stv + b3;                                       // (A^T p)[b3]
  TQ cur[4], nxt[4];
  // entry h of four values, as a two-level select on the bits of h (written as nested comparisons with h the compiler turns the
  // polish variant into EXEC-masked branches: some 80 instructions per stage for two of these)
  const bool hb0 = (h & 1) != 0, hb1 = (h & 2) != 0;
  auto by_h = [&](TQ a0, TQ a1, TQ a2, TQ a3) { const TQ lo = hb0 ? a1 : a0, hi = hb0 ? a3 : a2; return hb1 ? hi : lo; };
  const int zl = polish ? lane_zero() : 0;   // see lane_zero(): the pinned-input logic below stays on the vector ALU
  // affine: pad column 14 of the operand carries the gap (S[L.Dx], prepared by the caller), so that P c comes out of the
  // product P AB'' for free
  auto with_gap = [&](int st, TQ (&x)[4]) {
    TQ gq[4];
    vl_load(S + L.Dx + st * VS, h, gq);
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = vl ? gq[s] : x[s];
  };
  km.load(A, first, cur);
  if (affine) with_gap(first, cur);
  __syncthreads();
#pragma unroll MPCQ_UNROLL_FACTOR
  for (int i = first; i >= 0; --i) {
    PF_FAC(15);                          // (loop overhead, P update tail of the previous stage)
    km.load(A, i > 0 ? i - 1 : 0, nxt);   // a factorisation stage is long enough to hide one global fetch
    if (affine) with_gap(i > 0 ? i - 1 : 0, nxt);
    TQ qvi = 0;
    if (affine) qvi = A[L.qv + i * VS + b3];   // consumed at the end of the stage
    // what the solve needs and does not depend on this stage's products: read now, behind the tile products
    TQ rtv[4], rhov[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { rtv[a] = S[L.rt + i * NU + a + zl]; rhov[a] = S[L.rho + i * NU + a + zl]; }
    const TQ rtj = polish ? S[L.rt + i * NU + vj] : TQ(0);   // the Lambda^-1 lanes: is the input whose feed-forward they store pinned
    TQ acc1[4] = {0, 0, 0, 0}, acc2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma(acc1, Pop[s], cur[s]);              // T1''
    // column 14 of T1'' is P c_i (zero without the gap column): the vector pushed through AB''^T is p + P c
    TQ b2[4], vs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { vs[s] = affine ? pv[s] + acc1[s] : pv[s]; b2[s] = vl ? vs[s] : acc1[s]; }
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma(acc2, cur[s], b2[s]);               // F'' ; column 14 = AB''^T p
    PF_FAC(11);                          // two tile products (8 MFMA)
    // The ten entries of the stage Hessian come straight out of the accumulator tile through v_readlane (F''[10+a][10+q] is
    // register in_s(a) of lane (in_h(a), 10+q)): the LDL^T starts on them while the LDS round trip of everything else -- needed
    // only behind the factorisation -- is still in flight.
    TQ Lm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q <= a; ++q) Lm[a][q] = bc(acc2[in_s<TQ>(a)], 16 * in_h<TQ>(a) + 10 + q);
    if constexpr (polish && sizeof(TQ) == 4) {   // mixed precision: the curvature B'PB of every input on this working set (polish_mixed's report)
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) S[L.curv + i * NU + a] = Lm[a][a];
      }
    }
    // hand rows 10..13 over to the stage-Hessian lanes through LDS
    if (sizeof(TQ) == 4) {
      if (h >= 2) {   // h = 2: registers 2,3 = rows 10,11 ; h = 3: registers 0,1 = rows 12,13
        const TQ f0 = h == 2 ? acc2[2] : acc2[0], f1 = h == 2 ? acc2[3] : acc2[1];
        const TQ t0 = h == 2 ? acc1[2] : acc1[0], t1 = acc1[3];
        const int r0 = h == 2 ? 0 : 2;
        S[L.sF + r0 * VS + c] = f0;
        S[L.sF + (r0 + 1) * VS + c] = f1;
        S[L.sT + r0 * VS + c] = t0;
        if (h == 2) S[L.sT + VS + c] = t1;
      }
    } else {          // row 10 + j is register in_s(j) of lane group in_h(j)
      const TQ fv = h >= 2 ? acc2[2] : acc2[3], tvv = h >= 2 ? acc1[2] : acc1[3];
      const int rr = h >= 2 ? h - 2 : h + 2;
      S[L.sF + rr * VS + c] = fv;
      S[L.sT + rr * VS + c] = tvv;   // rr == 3: the spare row
    }
    if (vl) {
      TQ idp[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) idp[s] = sel.P[s] * vs[s];   // identity columns: (A^T v)[p] = v[p]
      vl_store(S + L.stv, h, idp);
      vl_store(S + L.stv + VS, h, acc2);                         // rows 0..9: A^T p ; rows 10..13: B^T p
    }
    __syncthreads();
    // ---- everything the solve reads from the hand-over, as ONE batch of LDS reads (no branch in between: the compiler
    //      issues them back to back and the LDL^T starts on the first arrivals)
    TQ tT[4], mvv[4], gu[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mvv[j] = S[m3off + j * m3str];                                 // M[j][c]
#pragma unroll
    for (int j = 0; j < 4; ++j) gu[j] = rhov[j] + S[L.stv + VS + 10 + j + zl];                   // gt = rho + B^T p
    const TQ tb = S[tboff];
#pragma unroll
    for (int s = 0; s < 4; ++s) tT[s] = S[toff[s]];   // transposed T1'' elements of the position columns of P_i (P update)
    // ---- Lambda = R~ + F_uu, LDL^T in registers (redundantly on every lane, straight-line code), then the solves
    TQ kk, mop, pcol;
    bool pin[4], pinany = false;             // input a of this stage is pinned (the same on every lane)
    {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        pin[a] = polish && rtv[a] < TQ(0);   // pinned input: diagonal pin_diag (see above), zero right-hand side of the feed-forward
        pinany = pinany || pin[a];
        Lm[a][a] += polish ? tabs(rtv[a]) : rtv[a];
      }
      PF_FAC(12);                        // LDS hand-over + operand reads
      // LDL^T (no square roots; the reciprocal pivots are the only long-latency operations of the chain):
      // Lm[a][q] (a > q) becomes the unit-lower factor, cm the unscaled column entries l*d.  (An explicit inverse by 2x2 blocks --
      // two reciprocals in series instead of four -- was measured in rounds 2 and 3: 1.3 % slower in fp64, and the fp32 mode
      // loses two orders of magnitude in accuracy with it.)
      TQ id[4], cm[4][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        TQ d = Lm[cc][cc];
#pragma unroll
        for (int k = 0; k < cc; ++k) d -= Lm[cc][k] * cm[cc][k];
        if (!(d > TQ(0))) ok = false;
        d = d > TQ(0) ? d : TQ(1);
        id[cc] = trcp1(d);
#pragma unroll
        for (int a = cc + 1; a < 4; ++a) {
          TQ s2 = Lm[a][cc];
#pragma unroll
          for (int k = 0; k < cc; ++k) s2 -= Lm[a][k] * cm[cc][k];
          cm[a][cc] = s2;
          Lm[a][cc] = s2 * id[cc];
        }
      }
      PF_FAC(13);                        // 4x4 LDL^T
      // rhs: M[:,c] on the lanes of column c < 13, e_vj on the four Lambda^-1 lanes
      TQ y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = c < NX ? mvv[j] : ((inv && vj == j) ? TQ(1) : TQ(0));
#pragma unroll
      for (int cc = 1; cc < 4; ++cc) {
#pragma unroll
        for (int k = 0; k < cc; ++k) y[cc] -= Lm[cc][k] * y[k];
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) y[cc] *= id[cc];
#pragma unroll
      for (int cc = 2; cc >= 0; --cc) {
#pragma unroll
        for (int k = cc + 1; k < 4; ++k) y[cc] -= Lm[k][cc] * y[k];
      }
      // operands of the P update first: they head the chain into the next stage
      const TQ yh = by_h(y[0], y[1], y[2], y[3]);
      const TQ mh = by_h(mvv[0], mvv[1], mvv[2], mvv[3]);
      kk = c < NX ? -yh : TQ(0);
      mop = mmask * mh;
      // column lanes: K[:,c] = -y, p_i[c] = (A^T p)[c] - y.gt (+ q_i[c]); Lambda^-1 lanes: row vj = y, k_vj = -y.gt
      TQ dot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const TQ gj = pin[j] ? TQ(0) : gu[j];
        dot += y[j] * gj;
        if (affine) gmax = tmax(gmax, tabs(gj));
      }
      TQ ex = tb - dot;
      if (affine) ex += qvi;                // stage gradient q_i enters the recursion directly
      pcol = c < NX ? ex : TQ(0);
      Kb[L.K + i * KS + h * ABW + c] = kk;   // K[h][c] from the lane that holds it; columns 13..15 are the zero pad of the row operands
      if (inv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Kb[L.Linv + i * 16 + vj * 4 + j] = y[j];
        S[L.vin + i * VS + vj] = (polish && rtj < TQ(0)) ? TQ(0) : -dot;
      }
    }
    PF_FAC(14);                          // right-hand sides, substitutions, stores of K, Lambda^-1
    // ---- P_i = Q + G + M^T K as one k=4 tile on top of the assembled C operand; p_i back to column 14 through DPP
    //      (also at i = 0, where nobody uses it: one tile product per factorisation instead of a branch in the chain)
    {
      TQ C4[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        C4[s] = mA2[s] * acc2[s] + mA1[s] * acc1[s] + mPo[s] * Pop[s] + mT[s] * tT[s] + qdg[s];
      mfma(C4, mop, kk);
#pragma unroll
      for (int s = 0; s < 4; ++s) { Pop[s] = C4[s]; cur[s] = nxt[s]; }
      l2g<TQ>(pcol, h, pv);
    }
    // ---- behind the chain: what the multiplier of a pinned input j needs [M_j | F_uu row j | gt_j], the cost-to-go tile
    if (affine && mrows && pinany) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!pin[j]) continue;
        if (h == 0 && c < NX) mrows[(i * NU + j) * MROW + c] = mvv[j];
        else if (inv && vj == j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) mrows[(i * NU + j) * MROW + NX + q] = S[L.sF + j * VS + 10 + q];
          mrows[(i * NU + j) * MROW + NX + 4] = gu[j];
        }
      }
    }
    if (i == 0) break;
    if (pstore && i <= pst_store && (i & (PSTEP - 1)) == 1) {   // cost-to-go of this stage, for a later restart below it
#pragma unroll
      for (int s = 0; s < 4; ++s) pstore[(i / PSTEP) * PST + lane * 4 + s] = Pop[s];
      if (vl) {
#pragma unroll
        for (int s = 0; s < 4; ++s) pstore[(i / PSTEP) * PST + 256 + RI<TQ>(s, h)] = pv[s];
      }
    }
    __syncthreads();   // the hand-over arrays are rewritten by the next stage
  }
  __syncthreads();
  if (affine && gscale) *gscale = gmax;   // every lane saw every stage gradient
  return wave_min<int>(ok ? 1 : 0) != 0;   // all lanes agree on definiteness
}

#ifdef MPCQ_DUMP_AT
template <typename TQ, typename PT> __device__ inline void dbg_dump(const DevModel<TQ>& m, int slot, PT base, int off, int n) {
  if (!m.dbg) return;
  double* d = m.dbg + (size_t)blockIdx.x * 4096 + slot;
  for (int i = lane_id(); i < n; i += 64) d[i] = (double)base[off + i];
}
#define DBG_DUMP(at, slot, base, off, n) do { if (MPCQ_DUMP_AT == (at)) dbg_dump(m, slot, base, off, n); } while (0)
#else
#define DBG_DUMP(at, slot, base, off, n)
#endif
// ------------------------------------------------------------------ QP: IPM + polish
// Mehrotra predictor-corrector iterations; every Newton system is one Riccati factorisation + two
// vector sweeps.  Continues from the current (z, sl, su, ll, lu, dx, grad) until |r_d| <= tol*gm and
// mu <= tol.  returns 0 converged / 1 NaN / 2 iteration cap / 4 stage Hessian not positive definite
// Shape-specialised instances with at most two inputs per lane (N <= 32; with four -- N = 50 -- the registers cost more than the
// passes: -1.3 %): the per-input quantities of an iteration (slacks, multipliers, their reciprocals, gradient, the two directions)
// stay in registers from one elementwise pass to the next instead of being re-read from LDS and re-derived in each of the seven
// passes -- the same expressions on the same operands, 4 reciprocals per input and iteration instead of 16.  (The elementwise passes
// were an eighth of an interior-point iteration: one wavefront issues an instruction every 4-5 cycles whatever it is.)  The LDS
// copies are kept current: the sweeps read rho and R~, the polish behind reads everything.
// The gradient follows an interior-point step without an adjoint sweep (H dz = -rho - Sigma dz elementwise) in both precisions.  In float
// the residual of the float solve piles up in it (which is why it was re-swept every iteration until round 5) -- but all the mixed-precision
// method needs from the interior point is the working set, which it then checks against double residuals: one sweep of four less per
// iteration, +2 % on the lockstep rate, the soaks as clean as before (-DMPCQ_F32_IPM_INCR=0: the gradient sweep per iteration).
#ifndef MPCQ_F32_IPM_INCR
#define MPCQ_F32_IPM_INCR 1
#endif
template <typename TQ> __device__ constexpr bool IPM_INCR() { return sizeof(TQ) == 8 || MPCQ_F32_IPM_INCR != 0; }
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB, typename M = DevModel<TQ>, typename PA = P<TQ>>
MPCQ_COLD int ipm_run_regs(const M& m, P<TQ> S, PA A, P<TQ> Kb, const Lds& L, const TQ tol, const TQ gm, int& it PF_ARG, const TQ rd_floor = TQ(3e-4), const int cap = 0) {
  constexpr int N = C::N > 0 ? C::N : 1, nv = N * NU, R = (nv + 63) / 64;
  const int tid = lane_id();
  int status = 2;
  const int maxit = (cap > 0 && cap < m.qp_max_iter) ? cap : m.qp_max_iter;
  bool on[R];
  int ix[R], gi[R];
  TQ sl[R], su[R], ll[R], lu[R], g[R], rr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + 64 * r;
    on[r] = i < nv; ix[r] = on[r] ? i : 0; gi[r] = GI(ix[r]);
    sl[r] = S[L.sl + ix[r]]; su[r] = S[L.su + ix[r]]; ll[r] = S[L.ll + ix[r]]; lu[r] = S[L.lu + ix[r]];
    g[r] = S[L.grad + gi[r]]; rr[r] = S[L.wq + 2 * VS + (ix[r] & 3)];
  }
  for (; it < maxit; ++it) {
    TQ rdm = 0, mu = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!on[r]) continue;
      rdm = tmax(rdm, tabs(g[r] - ll[r] + lu[r]));
      mu += sl[r] * ll[r] + su[r] * lu[r];
    }
    rdm = wave_max(rdm);
    mu = wave_sum(mu) / (2 * nv);
    if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("  ipm it %2d: |r_d|/gm %.3e  mu %.3e  (tol %.1e)\n", it, (double)(rdm / gm), (double)mu, (double)tol);
#endif
    // (float: the dual residual comes from a float gradient sweep whose own noise is ~1e-4 of the gradient scale on ill-conditioned
    //  instances -- asked for less, the iteration oscillates around that floor until its cap (1 solve in 3e5 at N = 50); what the
    //  active-set method behind needs from here is the complementarity, the residual it evaluates itself in double)
    if (rdm <= (sizeof(TQ) == 4 ? tmax(tol, rd_floor) : tol) * gm && mu <= tol) { status = 0; break; }
    // predictor: (H + Sigma) dza = -grad
    TQ rsl[R], rsu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      rsl[r] = trcp(sl[r]); rsu[r] = trcp(su[r]);
      if (on[r]) { S[L.rho + ix[r]] = g[r]; S[L.rt + ix[r]] = rr[r] + ll[r] * rsl[r] + lu[r] * rsu[r]; }
    }
    __syncthreads();
    PF_START();
    const bool fok = riccati_factor<C, false, false, TQ, GAB>(m, S, A, Kb, L PF_PASS, (TQ*)nullptr, P<TQ>(nullptr), P<TQ>(nullptr), -1, true);
    PF_STOP(PF_FACTOR);
    if (it == 0) { DBG_DUMP(2, 0, S, L.rt, nv); DBG_DUMP(2, 128, Kb, L.K, N * KS); DBG_DUMP(2, 2048, Kb, L.Linv, N * 16); DBG_DUMP(2, 3000, S, L.vin, N * VS); }
    if (!fok) { status = 4; break; }
    PF_START(); riccati_forward<C, false, TQ, GAB>(m, S, A, Kb, L, L.dza PF_PASS); PF_STOP(PF_FWD);
    if (it == 0) { DBG_DUMP(3, 0, S, L.dza, nv); DBG_DUMP(3, 128, S, L.Dx, (N + 1) * VS); }
    // step lengths without divisions: alpha = 1 / max_i(-ds_i / s_i); every quotient is a product with a reciprocal
    TQ da[R], dla[R], dua[R], rll[R], rlu[R], ainv = 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      da[r] = S[L.dza + ix[r]];
      rll[r] = trcp(ll[r]); rlu[r] = trcp(lu[r]);
      dla[r] = -ll[r] - ll[r] * rsl[r] * da[r]; dua[r] = -lu[r] + lu[r] * rsu[r] * da[r];
      if (on[r]) ainv = tmax(ainv, tmax(tmax(-da[r] * rsl[r], da[r] * rsu[r]), tmax(-dla[r] * rll[r], -dua[r] * rlu[r])));
    }
    const TQ aff = trcp(wave_max(ainv));
    TQ mua = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (on[r]) mua += (sl[r] + aff * da[r]) * (ll[r] + aff * dla[r]) + (su[r] - aff * da[r]) * (lu[r] + aff * dua[r]);
    mua = wave_sum(mua) / (2 * nv);
    TQ sigma = mua / mu;
    sigma = sigma * sigma * sigma;
    // float: the centring target does not go below a tenth of the tolerance -- from a warm start the complementarity can collapse (x 0.1
    // per iteration) while the residual lags, and at 1e-8 the float factorisation loses a stage Hessian's definiteness (1 solve in 1e6)
    const TQ smu = sizeof(TQ) == 4 ? tmax(sigma * mu, TQ(0.1) * tol) : sigma * mu;
    // corrector rhs r = -rd + rcl/sl - rcu/su ; linear term rho = -r
    TQ rcl[R], rcu[R], rho[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      rcl[r] = -sl[r] * ll[r] + smu - da[r] * dla[r];
      rcu[r] = -su[r] * lu[r] + smu + da[r] * dua[r];
      const TQ rd = g[r] - ll[r] + lu[r];
      rho[r] = rd - rcl[r] * rsl[r] + rcu[r] * rsu[r];
      if (on[r]) S[L.rho + ix[r]] = rho[r];
    }
    __syncthreads();
    PF_START(); riccati_backward_vec<C, TQ, GAB>(m, S, A, Kb, L, false); PF_STOP(PF_BWD);
    if (it == 0) { DBG_DUMP(4, 0, S, L.rho, nv); DBG_DUMP(4, 128, S, L.vin, N * VS); }
    PF_START(); riccati_forward<C, false, TQ, GAB>(m, S, A, Kb, L, L.dz PF_PASS); PF_STOP(PF_FWD);
    if (it == 0) { DBG_DUMP(5, 0, S, L.dz, nv); DBG_DUMP(5, 128, S, L.Dx, (N + 1) * VS); }
    TQ d[R], dl[R], du[R], apinv = 1, adinv = 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      d[r] = S[L.dz + ix[r]];
      dl[r] = (rcl[r] - ll[r] * d[r]) * rsl[r]; du[r] = (rcu[r] + lu[r] * d[r]) * rsu[r];
      if (on[r]) {
        apinv = tmax(apinv, tmax(-d[r] * rsl[r], d[r] * rsu[r]));
        adinv = tmax(adinv, tmax(-dl[r] * rll[r], -du[r] * rlu[r]));
      }
    }
    const TQ tau = tmax(TQ(0.995), 1 - mu);
    // alpha = min(1, tau / max_i(...)): the maxima start at 1, i.e. an unrestricted step has length tau
    TQ ap = tmin(TQ(1), tau * trcp(wave_max(apinv))), ad = tmin(TQ(1), tau * trcp(wave_max(adinv)));
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("             affine step %.3f  sigma %.2e  alpha_p %.4f  alpha_d %.4f\n", (double)aff, (double)sigma, (double)ap, (double)ad);
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // fp64: the gradient follows the step without a sweep (see ipm_run)
      if (IPM_INCR<TQ>()) g[r] += ap * (-rho[r] - (ll[r] * rsl[r] + lu[r] * rsu[r]) * d[r]);
      sl[r] = sl[r] + ap * d[r]; su[r] = su[r] - ap * d[r];
      ll[r] = ll[r] + ad * dl[r]; lu[r] = lu[r] + ad * du[r];
      if (on[r]) {
        S[L.z + ix[r]] += ap * d[r]; S[L.sl + ix[r]] = sl[r]; S[L.su + ix[r]] = su[r];
        S[L.ll + ix[r]] = ll[r]; S[L.lu + ix[r]] = lu[r];
        if (IPM_INCR<TQ>()) S[L.grad + gi[r]] = g[r];
      }
    }
    for (int i = tid; i < (N + 1) * VS; i += 64) S[L.dx + i] += ap * S[L.Dx + i];
    __syncthreads();
    if constexpr (!IPM_INCR<TQ>()) {
      PF_START(); adjoint<C, TQ, GAB>(m, S, A, L); PF_STOP(PF_ADJ);
#pragma unroll
      for (int r = 0; r < R; ++r) g[r] = S[L.grad + gi[r]];
    }
  }
  return status;
}
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_COLD int ipm_run(const DevModel<TQ>& m, P<TQ> S, P<TQ> A, P<TQ> Kb, const Lds& L, const TQ tol, const TQ gm, int& it PF_ARG) {
  if constexpr (C::N > 0 && C::N * NU <= 128) return ipm_run_regs<C>(m, S, A, Kb, L, tol, gm, it PF_PASS);
  const int N = cN<C>(m), nv = N * NU, tid = lane_id();
  int status = 2;
  const int maxit = m.qp_max_iter;
  for (; it < maxit; ++it) {
    TQ rdm = 0, mu = 0;
    for (int i = tid; i < nv; i += 64) {
      rdm = tmax(rdm, tabs(S[L.grad + GI(i)] - S[L.ll + i] + S[L.lu + i]));
      mu += S[L.sl + i] * S[L.ll + i] + S[L.su + i] * S[L.lu + i];
    }
    rdm = wave_max(rdm);
    mu = wave_sum(mu) / (2 * nv);
    if (!(rdm == rdm) || !(mu == mu)) { status = 1; break; }
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("  ipm it %2d: |r_d|/gm %.3e  mu %.3e  (tol %.1e)\n", it, (double)(rdm / gm), (double)mu, (double)tol);
#endif
    // (float: the dual residual comes from a float gradient sweep whose own noise is ~1e-4 of the gradient scale on ill-conditioned
    //  instances -- asked for less, the iteration oscillates around that floor until its cap (1 solve in 3e5 at N = 50); what the
    //  active-set method behind needs from here is the complementarity, the residual it evaluates itself in double)
    if (rdm <= (sizeof(TQ) == 4 ? tmax(tol, TQ(3e-4)) : tol) * gm && mu <= tol) { status = 0; break; }
    // predictor: (H + Sigma) dza = -grad
    for (int i = tid; i < nv; i += 64) S[L.rho + i] = S[L.grad + GI(i)];
    __syncthreads();
    PF_START();
    const bool fok = riccati_factor<C, false>(m, S, A, Kb, L PF_PASS);
    PF_STOP(PF_FACTOR);
    if (it == 0) { DBG_DUMP(2, 0, S, L.rt, nv); DBG_DUMP(2, 128, Kb, L.K, N * KS); DBG_DUMP(2, 2048, Kb, L.Linv, N * 16); DBG_DUMP(2, 3000, S, L.vin, N * VS); }
    if (!fok) { status = 4; break; }
    PF_START(); riccati_forward<C>(m, S, A, Kb, L, L.dza PF_PASS); PF_STOP(PF_FWD);
    if (it == 0) { DBG_DUMP(3, 0, S, L.dza, nv); DBG_DUMP(3, 128, S, L.Dx, (N + 1) * VS); }
    // Step lengths without divisions: alpha = min_i(-s_i / ds_i | ds_i < 0) = 1 / max_i(-ds_i / s_i), and every quotient by a
    // slack or a multiplier is a product with its reciprocal (v_rcp_f64 + two Newton steps instead of the ~35-instruction
    // IEEE division, 28 of which per input and iteration were 8 % of an iteration).
    TQ ainv = 1;
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ rsl = trcp(sl), rsu = trcp(su);
      const TQ dl = -ll - ll * rsl * d, du = -lu + lu * rsu * d;
      ainv = tmax(ainv, tmax(tmax(-d * rsl, d * rsu), tmax(-dl * trcp(ll), -du * trcp(lu))));
    }
    const TQ aff = trcp(wave_max(ainv));
    TQ mua = 0;
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dl = -ll - ll * trcp(sl) * d, du = -lu + lu * trcp(su) * d;
      mua += (sl + aff * d) * (ll + aff * dl) + (su - aff * d) * (lu + aff * du);
    }
    mua = wave_sum(mua) / (2 * nv);
    TQ sigma = mua / mu;
    sigma = sigma * sigma * sigma;
    const TQ smu = sizeof(TQ) == 4 ? tmax(sigma * mu, TQ(0.1) * tol) : sigma * mu;   // (see ipm_run_regs)
    // corrector rhs r = -rd + rcl/sl - rcu/su ; linear term rho = -r
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dza + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ rsl = trcp(sl), rsu = trcp(su);
      const TQ dl = -ll - ll * rsl * d, du = -lu + lu * rsu * d;
      const TQ rcl = -sl * ll + smu - d * dl;
      const TQ rcu = -su * lu + smu + d * du;
      const TQ rd = S[L.grad + GI(i)] - ll + lu;
      S[L.rho + i] = rd - rcl * rsl + rcu * rsu;
    }
    __syncthreads();
    PF_START(); riccati_backward_vec<C>(m, S, A, Kb, L, false); PF_STOP(PF_BWD);
    if (it == 0) { DBG_DUMP(4, 0, S, L.rho, nv); DBG_DUMP(4, 128, S, L.vin, N * VS); }
    PF_START(); riccati_forward<C>(m, S, A, Kb, L, L.dz PF_PASS); PF_STOP(PF_FWD);
    if (it == 0) { DBG_DUMP(5, 0, S, L.dz, nv); DBG_DUMP(5, 128, S, L.Dx, (N + 1) * VS); }
    TQ apinv = 1, adinv = 1;
    for (int i = tid; i < nv; i += 64) {
      const TQ da = S[L.dza + i], d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ rsl = trcp(sl), rsu = trcp(su);
      const TQ dla = -ll - ll * rsl * da, dua = -lu + lu * rsu * da;
      const TQ rcl = -sl * ll + smu - da * dla;
      const TQ rcu = -su * lu + smu + da * dua;
      const TQ dl = (rcl - ll * d) * rsl, du = (rcu + lu * d) * rsu;
      apinv = tmax(apinv, tmax(-d * rsl, d * rsu));
      adinv = tmax(adinv, tmax(-dl * trcp(ll), -du * trcp(lu)));
      S[L.act + i] = dl; S[L.rt + i] = du;   // kept for the update below (neither array is live inside an interior-point iteration here)
    }
    const TQ tau = tmax(TQ(0.995), 1 - mu);
    // alpha = min(1, tau / max_i(...)): the maxima start at 1, i.e. an unrestricted step has length tau
    TQ ap = tmin(TQ(1), tau * trcp(wave_max(apinv))), ad = tmin(TQ(1), tau * trcp(wave_max(adinv)));
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("             affine step %.3f  sigma %.2e  alpha_p %.4f  alpha_d %.4f\n", (double)aff, (double)sigma, (double)ap, (double)ad);
#endif
    for (int i = tid; i < nv; i += 64) {
      const TQ d = S[L.dz + i], sl = S[L.sl + i], su = S[L.su + i], ll = S[L.ll + i], lu = S[L.lu + i];
      const TQ dl = S[L.act + i], du = S[L.rt + i];
      S[L.z + i] += ap * d; S[L.sl + i] = sl + ap * d; S[L.su + i] = su - ap * d;
      S[L.ll + i] = ll + ad * dl; S[L.lu + i] = lu + ad * du;
      // fp64: the gradient follows the step without a sweep.  The corrector solved (H + Sigma) dz = -rho with
      // Sigma = ll/sl + lu/su of this iteration, so H dz = -rho - Sigma dz elementwise (the solve leaves a residual at the
      // rounding level of double, far below the hand-over tolerance; the active-set iterations that follow recompute
      // everything).  fp32 keeps the adjoint sweep: there the residual of the solve would pile up in the gradient.
      if (IPM_INCR<TQ>()) S[L.grad + GI(i)] += ap * (-S[L.rho + i] - (ll * trcp(sl) + lu * trcp(su)) * d);
    }
    for (int i = tid; i < (N + 1) * VS; i += 64) S[L.dx + i] += ap * S[L.Dx + i];
    __syncthreads();
    if (!IPM_INCR<TQ>()) { PF_START(); adjoint<C>(m, S, A, L); PF_STOP(PF_ADJ); }
  }
  return status;
}

// Active-set polish: starting from the IPM point, pin the inputs the IPM identifies as active and
// take Newton steps on the free set (masked Riccati) with a ratio test; inputs whose multiplier has
// the wrong sign are released (the worst one, only at a minimiser of the current working set), inputs
// that block are pinned.  Ends on an exact KKT point of the QP (to rounding), which an interior
// method only approaches like sqrt(mu) on weakly active bounds.
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_PHASE bool polish(const DevModel<TQ>& m, P<TQ> S, P<TQ> A, P<TQ> G, const Lds& L, TQ gm, int& passes, const bool warm, const int max_passes, int& why PF_ARG) {
  why = QPX_BUDGET;
  const int N = cN<C>(m), nv = N * NU, tid = lane_id();
  const P<TQ> Kb = C::GK ? G : S;   // where the gains live
  PF_MARK(12);                      // (12: working-set set-up of a pass, up to its factorisation)
  if (warm) {   // working set = inputs the previous iterate left exactly on a bound; start from z = 0 (feasible)
    for (int i = tid; i < nv; i += 64) {
      S[L.act + i] = S[L.lb + i] == TQ(0) ? TQ(-1) : (S[L.ub + i] == TQ(0) ? TQ(1) : TQ(0));
      S[L.z + i] = 0;
    }
  } else {      // working set identified by the interior point
    for (int i = tid; i < nv; i += 64)
      S[L.act + i] = S[L.ll + i] > m.pin_ratio * S[L.sl + i] ? TQ(-1) : (S[L.lu + i] > m.pin_ratio * S[L.su + i] ? TQ(1) : TQ(0));
  }
  __syncthreads();
  TQ tolm = (sizeof(TQ) == 4 ? TQ(8) : TQ(64)) * m.eps * gm;  // multiplier sign test
  const TQ tolb = 16 * m.eps;       // bound proximity (bounds are O(1))
  bool settled = false;
  int nact = 1;   // pinned inputs in the working set
  // a bulk release that bounces straight back (the freed inputs violate and get pinned again) makes the next ones
  // more selective: all wrong-signed multipliers -> those within 4x of the worst -> within 1.6x -> the worst only
  int careful = 0;
  bool released = false;
  unsigned relmask = 0;   // bit k: input tid + 64 k was released in the previous pass (a bounce = it is pinned again in this one)
  int top = N - 1;   // highest stage whose working set changed since the last factorisation (N-1: factorise everything)
  bool keep_p = false;
  int ptop = -1;     // highest stage whose working set may change (tiles are kept for stages <= ptop + 1)
  int pst_hi = -1;   // highest stage whose stored cost-to-go tile is current
  for (passes = 0; passes < max_passes; ++passes) {
    PF_MARK(12);
    {
      // New working set.  Its minimiser is the solution of the affine LQ problem with the pinned inputs held at their
      // bounds: their effect B_i zbar_i joins the gap c_i, the stage gradients q_i, r_i enter the vector recursion
      // directly -- ONE masked factorisation and ONE forward sweep, whatever the current point is (no state rollout,
      // no gradient sweep first).
      int na = 0;
      for (int i = tid; i < nv; i += 64) {
        const TQ a = S[L.act + i];
        if (a < 0) S[L.z + i] = S[L.lb + i];
        else if (a > 0) S[L.z + i] = S[L.ub + i];
        S[L.rho + i] = S[L.r0 + i];
        na += a != TQ(0) ? 1 : 0;
      }
      nact = wave_sum(na);
#ifdef MPCQ_EMU_DEBUG
      if (passes == 0) {
        for (int i = tid; i < nv; i += 64)
          if (S[L.act + i] != TQ(0)) printf("       start: pinned stage %2d rotor %d %s\n", i >> 2, i & 3, S[L.act + i] < 0 ? "lower" : "upper");
      }
#endif
      if (passes == 0) {
        // Cost-to-go tiles are kept (43 KB of global stores) only where a change of the working set is likely: some
        // input pinned already, or a free one within `margin` of a bound.  Elsewhere a change (rare) refactorises from
        // the top as before.
        // ... and only up to the highest such stage: the recursion runs from the last stage down, a restart at stage t needs the tile
        // of stage t + 1, and above the highest stage that can change nothing is ever restarted (a change up there, rare, refactorises
        // from the top and extends the range).  On the bench workload the saturated inputs sit in the first stages of the horizon.
        int hi = -1;
        for (int i = tid; i < nv; i += 64)
          if (S[L.act + i] != TQ(0) || tmin(-S[L.lb + i], S[L.ub + i]) < TQ(0.1)) hi = i >> 2;   // (ascending i per lane: the last hit is the highest)
        ptop = wave_max(hi);
        keep_p = ptop >= 0;
#ifdef MPCQ_AB_NO_PSTORE   // A/B measurement only: no cost-to-go tiles, every factorisation starts at the last stage
        keep_p = false;
#endif
      }
      __syncthreads();
#if MPCQ_G_DXBATCH
      // (the gaps come from the stage records -- global memory in the GAB layouts: the loads of a block are issued together, one memory
      //  round trip per block instead of one per 64 elements; the pad slots 13..15 of c are zero, shoot_sens)
      constexpr int CB = 5;
      for (int base = 0; base < N * VS; base += 64 * CB) {
        TQ cv[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) { const int it = base + 64 * u + tid; cv[u] = A[L.c + (it < N * VS ? it : 0)]; }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          const int it = base + 64 * u + tid, i = it >> 4, r = it & 15;
          if (it >= N * VS) continue;
          TQ v = cv[u];
          if (nact > 0 && r < NX) {
#pragma unroll
            for (int j = 0; j < NU; ++j)
              if (S[L.act + i * NU + j] != TQ(0)) v += A[L.AB + i * ABS + r * ABW + 10 + j] * S[L.z + i * NU + j];
          }
          S[L.Dx + it] = v;
        }
      }
#else
      for (int it = tid; it < N * VS; it += 64) {
        const int i = it >> 4, r = it & 15;
        TQ v = 0;
        if (r < NX) {
          v = A[L.c + it];
          if (nact > 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j)
              if (S[L.act + i * NU + j] != TQ(0)) v += A[L.AB + i * ABS + r * ABW + 10 + j] * S[L.z + i * NU + j];
          }
        }
        S[L.Dx + it] = v;
      }
#endif
      __syncthreads();
      PF_START();
      TQ gfac = 0;
#if defined(MPCQ_PROFILE) && !defined(MPCQ_PROFILE_FWD) && !defined(MPCQ_PROFILE_FAC) && !defined(MPCQ_PROFILE_SERIAL) && !defined(MPCQ_PROFILE_OTHER)
      pf.acc[13] += (keep_p ? pst_first(N, top, pst_hi) : N - 1) + 1;   // stages this factorisation visits
      pf.acc[14] += 1;                                                    // factorisations
#endif
      if (passes > 0 && top > ptop) ptop = top;   // a change above the range: this factorisation starts at the top (no tile there) and stores up to it
      const int pst_store = ((ptop + PSTEP - 1) & ~(PSTEP - 1)) + 1;   // the tile a restart at or below ptop can need
      const bool fok = riccati_factor<C, true, true>(m, S, A, Kb, L PF_PASS, &gfac, nact > 0 ? G + L.mrow : P<TQ>(nullptr), keep_p ? G + L.pst : P<TQ>(nullptr), top, false,
                                                     pst_hi, pst_store);
      if (!keep_p || pst_first(N, top, pst_hi) == N - 1) pst_hi = keep_p ? pst_store : -1;   // a full factorisation rewrote every tile it keeps; a resumed one left those above untouched
      top = -1;
      PF_STOP(PF_FACTOR);
      PF_BUCKET(13);                     // (13: between factorisation and forward sweep)
      if (!fok) { why = QPX_NUMERIC; return false; }
      gm = tmax(gm, tmax(TQ(1), gfac));   // a restarted factorisation sees only the stages it visits
      tolm = (sizeof(TQ) == 4 ? TQ(8) : TQ(64)) * m.eps * gm;
      PF_START(); riccati_forward<C, true>(m, S, A, Kb, L, L.dz PF_PASS); PF_STOP(PF_FWD);
      PF_BUCKET(14);                     // (14: ratio test, multipliers, step of a pass)
      // the sweep returns the minimiser itself: turn it into a step from the current point for the ratio test below
      for (int i = tid; i < nv; i += 64)
        if (S[L.act + i] == TQ(0)) S[L.dz + i] -= S[L.z + i];
      __syncthreads();
    }
    // Does the minimiser of the working set leave the box?  (ratio test of the step from the current feasible point:
    // alpha < 1  <=>  some free input ends outside its bounds; only that bit is needed, so no divisions)
    int flags = 0;   // bit 0: a free input violates a bound, bit 1: not a number
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const TQ zn = S[L.z + i] + S[L.dz + i];
      if (!(zn == zn)) flags |= 2;
      if (zn < S[L.lb + i] || zn > S[L.ub + i]) flags |= 1;
    }
    flags = wave_reduce(flags, [](int a, int b) { return a | b; });
    if (flags & 2) { why = QPX_NUMERIC; return false; }
    const bool feasible = !(flags & 1);
    // Multipliers of the pinned inputs at the minimiser of the working set, without a gradient sweep: with the
    // cost-to-go of the factorisation, lambda_a = gt_a + M_a dx_i + sum_{q free} (B'PB)_aq z_q + R_aa z_a (rows left
    // behind by riccati_factor).  A feasible minimiser with correctly signed multipliers is the solution.  Otherwise the
    // wrong-signed ones are released (only at a feasible minimiser: the classical primal rule).
    bool any_release = false;
    unsigned relnow = 0;
    if (nact > 0 && feasible) {
      TQ vmax = 0;
      for (int i = tid; i < nv; i += 64) {
        const TQ a = S[L.act + i];
        TQ lam = 0;
        if (a != TQ(0)) {
          const int st = i >> 2, j = i & 3;
          const P<TQ> row = G + (L.mrow + i * MROW);
          TQ rv[NX + 5];
#pragma unroll
          for (int k = 0; k < NX + 5; ++k) rv[k] = row[k];
          lam = rv[NX + 4] + S[L.wq + 2 * VS + j] * S[L.z + i];
#pragma unroll
          for (int k = 0; k < NX; ++k) lam += rv[k] * S[L.dx + st * VS + k];
#pragma unroll
          for (int q = 0; q < NU; ++q)
            if (q != j && S[L.act + st * NU + q] == TQ(0)) lam += rv[NX + q] * (S[L.z + st * NU + q] + S[L.dz + st * NU + q]);
          vmax = tmax(vmax, a < 0 ? -lam : lam);
        }
        S[L.grad + GI(i)] = lam;
      }
      vmax = wave_max(vmax);
      if (!(vmax == vmax)) { why = QPX_NUMERIC; return false; }
#ifdef MPCQ_EMU_DEBUG
      if (tid == 0) printf("  polish pass %d warm %d feasible %d vmax %.3e tolm %.3e nact %d\n", passes, (int)warm, (int)feasible, (double)vmax, (double)tolm, nact);
#endif
      if (vmax > tolm) {
        __syncthreads();
        // Which of the wrong-signed ones: the multipliers of one rotor's run of saturated stages are strongly coupled
        // (freeing the input at the end of the run turns the others' signs back), so a bulk release of the whole run is
        // followed by one re-pinning pass per stage.  Rule: per rotor only the worst multiplier goes (input i belongs to
        // rotor i & 3 = lane & 3: every lane sees one rotor); after bounces fewer rotors, in the end only the worst one overall.
        TQ vr = 0;
        int nwrong = 0;
        for (int i = tid; i < nv; i += 64) {
          const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
          if (a != TQ(0)) { const TQ v = a < 0 ? -g : g; vr = tmax(vr, v); nwrong += v > tolm ? 1 : 0; }
        }
        // many wrong-signed multipliers at once in a warm attempt: the releases would go rotor by rotor, pass after pass
        if (warm && m.abort_wrong > 0 && wave_sum(nwrong) >= m.abort_wrong) { why = QPX_WRONG; return false; }
        TQ rel_thr = vmax;   // careful == 3: the worst one overall (the classical rule)
        if (careful < 3) {   // per rotor the worst one; after bounces only rotors whose worst is within 4x / 1.6x of the overall worst
          const TQ w0 = wave_max((tid & 3) == 0 ? vr : TQ(0)), w1 = wave_max((tid & 3) == 1 ? vr : TQ(0)),
                   w2 = wave_max((tid & 3) == 2 ? vr : TQ(0)), w3 = wave_max((tid & 3) == 3 ? vr : TQ(0));
          rel_thr = (tid & 3) == 0 ? w0 : ((tid & 3) == 1 ? w1 : ((tid & 3) == 2 ? w2 : w3));
          rel_thr = tmax(rel_thr, careful == 0 ? TQ(0) : (careful == 1 ? TQ(0.25) * vmax : TQ(0.625) * vmax));
        }
        int hi = -1;
        for (int i = tid; i < nv; i += 64) {
          const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
          const TQ v = a < 0 ? -g : g;
          if (a != TQ(0) && v > tolm && v >= rel_thr) {   // release (they stay where they are)
#ifdef MPCQ_EMU_DEBUG
            printf("       release stage %2d rotor %d %s  multiplier %+.4e\n", i >> 2, i & 3, a < 0 ? "lower" : "upper", (double)g);
#endif
            S[L.act + i] = 0; S[L.dz + i] = 0; hi = i >> 2;
            relnow |= 1u << ((i >> 6) & 31);
          }
        }
        top = tmax(top, wave_max(hi));
        any_release = true;
      }
    }
    // full step; if it leaves the box, clip and pin EVERY violator at once (the minimiser on a working set does not
    // depend on the starting point, so only the sequence of working sets matters)
    int nblk = 0, hip = -1, bounce = 0;
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const TQ lb = S[L.lb + i], ub = S[L.ub + i];
      TQ z = S[L.z + i] + S[L.dz + i];
      if (!feasible) {
        const bool lo = z <= lb + tolb, up = !lo && z >= ub - tolb;
        if (lo || up) {
#ifdef MPCQ_EMU_DEBUG
          printf("       pin stage %2d rotor %d %s  (unclipped %+.4e, bound %+.4e, had z %+.4e)\n", i >> 2, i & 3, lo ? "lower" : "upper", (double)z, (double)(lo ? lb : ub), (double)S[L.z + i]);
#endif
          z = lo ? lb : ub; S[L.act + i] = lo ? TQ(-1) : TQ(1); nblk += 1; hip = i >> 2;
          bounce |= (relmask >> ((i >> 6) & 31)) & 1u;
        }
      }
      S[L.z + i] = z;
    }
    nblk = wave_sum(nblk);
    bounce = wave_max(bounce);
    top = tmax(top, wave_max(hip));
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("     pass %d feasible %d nblk %d release %d\n", passes, (int)feasible, nblk, (int)any_release);
#endif
#if defined(MPCQ_PROFILE) && !defined(MPCQ_PROFILE_FWD) && !defined(MPCQ_PROFILE_FAC) && !defined(MPCQ_PROFILE_SERIAL) && !defined(MPCQ_PROFILE_OTHER)
    pf.acc[11] += nblk;                        // inputs pinned
    pf.acc[12] += any_release ? 1 : 0;         // passes with a release
    pf.acc[15] += (any_release && nblk > 0) ? 1 : 0;   // passes with both
#endif
    if (feasible && !any_release) { settled = true; passes += 1; break; }
    // a bulk release that bounces back wholesale with most of the inputs saturated: far from the optimal working set,
    // the interior point gets there faster
    if (released && nblk >= 8 && 2 * (nact + nblk) >= nv) { why = QPX_BOUNCE; return false; }
    // the first minimiser of a warm attempt leaves the box in many inputs at once: the previous working set is no guess (the
    // saturated inputs moved to other rotors); passes would follow one another, the interior point gets there faster
    if (warm && passes == 0 && m.abort_pins > 0 && nblk >= m.abort_pins) { why = QPX_PINS; return false; }
    if (bounce && careful < 3) careful += 1;   // an input released in the previous pass is pinned again
    released = any_release;
    relmask = relnow;
    __syncthreads();
  }
  if (settled && sizeof(TQ) == 4) {   // f32: replace the incrementally updated trajectory by a fresh rollout of the final z
    PF_START(); rollout<C>(m, S, A, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
  }
  return settled;
}

// ------------------------------------------------------------------ mixed precision (TQ = float): fp64 residuals on the float stage records
// The float Riccati factorisation is a preconditioner: cond(H) ~ 2e6, so a solve in float alone carries 1e-5 .. 1e-2 of error in du
// (hard part 1b).  The solution is therefore kept in double (D[L.zd], D[L.dxd]) and refined against the residual
// of the QP evaluated in DOUBLE arithmetic on the stored (float) stage records -- state rollout and gradient sweep below, the same
// operand registers as the float sweeps with the float slot map, the vector in double -- with the float factorisation solving for
// the corrections: classical iterative refinement, contraction ~ cond x eps32 per step, limit = the exact solution of the QP the
// records define.
// state trajectory of D[L.zd]: dxd_{i+1} = A dxd_i + B zd_i + c_i from D[L.dxd + 0..15] = dx_0
template <typename C, bool GAB = C::GAB>
MPCQ_PHASE void rollout64(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, const Lds& L) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15;
  const RMaj<float> rm(L.AB + N * ABS, L.AB, ABS, NX, h, c);
  constexpr int PD = Depth<GAB>::PD;
  const bool prow = c >= 10 && c < NX;
  // float slot map: lane row h multiplies slots 4h .. 4h+3 -- states below 10, inputs 0,1 in registers 2,3 of row 2, inputs 2,3 in
  // registers 0,1 of row 3 (slots 14, 15 are the zero pad columns of AB'')
  const int zo = L.zd + (h == 3 ? 2 : 0);
  double xc = D[L.dxd + c];
  float qa[PD + 1][4], qc[PD + 1];
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int id = d < N ? d : N - 1;
    rm.load(A, id, qa[d]);
    qc[d] = A[L.c + id * VS + c];
  }
#pragma unroll MPCQ_UNROLL_SWEEP
  for (int i = 0; i < N; ++i) {
    const int ip = i + PD < N ? i + PD : N - 1;
    rm.load(A, ip, qa[PD]);
    qc[PD] = A[L.c + ip * VS + c];
    const double z0 = D[zo + i * NU], z1 = D[zo + i * NU + 1];
    double xv[4];
    l2g<float>(xc, h, xv);
    const double v0 = h == 3 ? z0 : xv[0], v1 = h == 3 ? z1 : xv[1];
    const double v2 = h < 2 ? xv[2] : (h == 2 ? z0 : 0.0), v3 = h < 2 ? xv[3] : (h == 2 ? z1 : 0.0);
    const double t = hsum(((double)qa[0][0] * v0 + (double)qa[0][1] * v1) + ((double)qa[0][2] * v2 + (double)qa[0][3] * v3));
    double xn = t + (prow ? xc : 0.0) + (double)qc[0];
    xn = c < NX ? xn : 0.0;
    xc = xn;
    if (lane < VS) D[L.dxd + (i + 1) * VS + lane] = xn;
    shift<float, PD>(qa);
#pragma unroll
    for (int d = 0; d < PD; ++d) qc[d] = qc[d + 1];
  }
  __syncthreads();
}
// gradient of the QP objective at (dxd(zd), zd) in double, rounded to float into S[L.grad] (the right-hand side of the correction
// solve; the multipliers of the pinned inputs); gF = largest |gradient| over the free inputs, vmax = worst wrong-signed multiplier
// of a pinned one, both from the double values
template <typename C, bool GAB = C::GAB>
MPCQ_PHASE void adjoint64(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, const Lds& L, double& gF, double& vmax) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15;
  const KMaj<float> km(L, N, h, c);
  constexpr int PD = Depth<GAB>::PD;
  const bool arow = c < 10, prow = c >= 10 && c < NX, ucol = c >= 10 && c < 14;
  const int j = ucol ? c - 10 : 0;
  const double qdc = (double)S[L.wq + c], ru = (double)S[L.wq + 2 * VS + j];
  double pc = (double)S[L.wq + VS + c] * D[L.dxd + N * VS + c] + (double)A[L.qv + N * VS + c];
  float qa[PD + 1][4], qq[PD + 1];
  double gf = 0, vm = 0;
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int id = N - 1 - d > 0 ? N - 1 - d : 0;
    km.load(A, id, qa[d]);
    qq[d] = A[L.qv + id * VS + c];
  }
#pragma unroll MPCQ_UNROLL_SWEEP
  for (int i = N - 1; i >= 0; --i) {
    const int ip = i - PD > 0 ? i - PD : 0;
    km.load(A, ip, qa[PD]);
    qq[PD] = A[L.qv + ip * VS + c];
    const double dxc = D[L.dxd + i * VS + c];
    // local part of the input gradient: R z + r0 (r0 = R (U - uref), formed in double by the load phase and rounded to float once)
    const double gvc = ucol ? ru * D[L.zd + i * NU + j] + (double)S[L.r0 + i * NU + j] : 0.0;
    const float a = S[L.act + i * NU + j];
    double pi[4];
    l2g<float>(pc, h, pi);
    const double t = hsum(((double)qa[0][0] * pi[0] + (double)qa[0][1] * pi[1]) + ((double)qa[0][2] * pi[2] + (double)qa[0][3] * pi[3]));   // (AB''^T p)[c]
    const double g = t + gvc;
    if (lane < VS) S[L.grad + i * VS + lane] = (float)g;
    if (ucol) {
      double ag = fabs(g);
      if (!(ag == ag)) ag = 1e308;   // not a number: reported as an overflowing residual (the caller tests gF < 1e300)
      if (a == 0.0f) gf = ag > gf ? ag : gf;
      else { const double v = a < 0.0f ? -g : g; vm = v > vm ? v : vm; gf = ag >= 1e308 ? ag : gf; }
    }
    pc = (arow ? t : (prow ? pc : 0.0)) + (qdc * dxc + (double)qq[0]);
    shift<float, PD>(qa);
#pragma unroll
    for (int d = 0; d < PD; ++d) qq[d] = qq[d + 1];
  }
  __syncthreads();
  gF = wave_max(gf);
  vmax = wave_max(vm);
}

// Forward sweep of the float gains with the vector in double (see riccati_forward for the lane roles):
//   affine:  z_i = K_i dx_i + k_i (pinned inputs: exactly their bound, 0 in a warm attempt), dx_{i+1} = A dx_i + B z_i + c_i from dx_0 = D[L.dxd + 0..15]
//            -> D[L.zd], D[L.dxd]: solution of the affine LQ problem on the working set and ITS OWN state rollout, consistent to double rounding;
//   !affine: dz_i = K_i Dx_i + k_i, Dx_{i+1} = A Dx_i + B dz_i from Dx_0 = 0 -> S[L.dz] (float) and D[L.dxd] += Dx: the trajectory follows
//            the correction inside the sweep (the rollout is linear), so no rollout64 is needed behind a refinement step.
template <typename C, bool affine, bool GAB = C::GAB>
MPCQ_PHASE void riccati_forward64(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, P<float> Kb, const Lds& L PF_ARG) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15;
  const RMaj<float> rm(L.AB + N * ABS, L.AB, ABS, NX, h, c);
  constexpr int PD = Depth<GAB>::PD;
  const bool prow = c >= 10 && c < NX;
  // float slot map: lane row 2 holds inputs 0, 1 in registers 2, 3, row 3 inputs 2, 3 in registers 0, 1; each of the two rows computes its own pair
  const int j0 = h == 3 ? 2 : 0, j1 = j0 + 1;
  const bool hasu = h >= 2;
  const int ko0 = L.K + j0 * ABW + c, ko1 = L.K + j1 * ABW + c;
  constexpr int KD = C::GK ? PD : 1;
  double xc = affine ? D[L.dxd + c] : 0.0;
  float qa[PD + 1][4], kq0[KD + 1], kq1[KD + 1], qc[PD + 1], g0, g1, p0 = 0, p1 = 0;
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int id = d < N ? d : N - 1;
    rm.load(A, id, qa[d]);
    qc[d] = affine ? A[L.c + id * VS + c] : 0.0f;
  }
#pragma unroll
  for (int d = 0; d < KD; ++d) {
    const int id = d < N ? d : N - 1;
    kq0[d] = Kb[ko0 + id * KS];
    kq1[d] = Kb[ko1 + id * KS];
  }
  g0 = S[L.vin + j0]; g1 = S[L.vin + j1];
  if (affine) { p0 = S[L.rt + j0]; p1 = S[L.rt + j1]; }
#pragma unroll MPCQ_UNROLL_SWEEP
  for (int i = 0; i < N; ++i) {
    const int ip = i + 1 < N ? i + 1 : i, ig = i + PD < N ? i + PD : N - 1, ik = i + KD < N ? i + KD : N - 1;
    rm.load(A, ig, qa[PD]);
    qc[PD] = affine ? A[L.c + ig * VS + c] : 0.0f;
    kq0[KD] = Kb[ko0 + ik * KS];
    kq1[KD] = Kb[ko1 + ik * KS];
    const float g0n = S[L.vin + ip * VS + j0], g1n = S[L.vin + ip * VS + j1];
    float p0n = 0, p1n = 0;
    if (affine) { p0n = S[L.rt + ip * NU + j0]; p1n = S[L.rt + ip * NU + j1]; }
    double u0 = rowsum((double)kq0[0] * xc) + (double)g0, u1 = rowsum((double)kq1[0] * xc) + (double)g1;
    if (affine) { u0 = p0 < 0.0f ? 0.0 : u0; u1 = p1 < 0.0f ? 0.0 : u1; }   // pinned (R~ < 0): held at the bound, exactly
    double xv[4];
    l2g<float>(xc, h, xv);
    const double v0 = h == 3 ? u0 : xv[0], v1 = h == 3 ? u1 : xv[1];
    const double v2 = h < 2 ? xv[2] : (h == 2 ? u0 : 0.0), v3 = h < 2 ? xv[3] : (h == 2 ? u1 : 0.0);
    const double ta = ((double)qa[0][0] * v0 + (double)qa[0][1] * v1) + ((double)qa[0][2] * v2 + (double)qa[0][3] * v3);
    if (c == 0 && hasu) {
      if (affine) { D[L.zd + i * NU + j0] = u0; D[L.zd + i * NU + j1] = u1; }
      else { S[L.dz + i * NU + j0] = (float)u0; S[L.dz + i * NU + j1] = (float)u1; }
    }
    double xn = hsum(ta) + (prow ? xc : 0.0) + (affine ? (double)qc[0] : 0.0);
    xn = c < NX ? xn : 0.0;
    xc = xn;
    if (lane < VS) {
      if (affine) D[L.dxd + (i + 1) * VS + lane] = xn;
      else D[L.dxd + (i + 1) * VS + lane] += xn;
    }
    g0 = g0n; g1 = g1n; p0 = p0n; p1 = p1n;
    shift<float, PD>(qa);
#pragma unroll
    for (int d = 0; d < PD; ++d) qc[d] = qc[d + 1];
#pragma unroll
    for (int d = 0; d < KD; ++d) { kq0[d] = kq0[d + 1]; kq1[d] = kq1[d + 1]; }
  }
  __syncthreads();
}
// adjoint64 and the backward vector recursion of the float factorisation (riccati_backward_vec, polish form) as ONE backward sweep: the
// gradient of stage i is the linear term of the recursion at stage i, both recursions read the same operand registers, and their two
// dependent chains interleave.  Used when the factorisation in K / Lambda^-1 belongs to the current working set.
template <typename C, bool GAB = C::GAB>
MPCQ_PHASE void adjoint64_bwd(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, P<float> Kb, const Lds& L, double& gF, double& vmax) {
  const int N = cN<C>(m), lane = lane_id(), h = lane >> 4, c = lane & 15;
  const KMaj<float> km(L, N, h, c);
  constexpr int PD = Depth<GAB>::PD;
  constexpr int KD = C::GK ? PD : 0;
  const bool arow = c < 10, prow = c >= 10 && c < NX, ucol = c >= 10 && c < 14;
  const int j = ucol ? c - 10 : 0, lj = lane < NU ? lane : 0;
  const double qdc = (double)S[L.wq + c], ru = (double)S[L.wq + 2 * VS + j];
  double pc = (double)S[L.wq + VS + c] * D[L.dxd + N * VS + c] + (double)A[L.qv + N * VS + c];   // costate of the objective
  float pr = 0;                                                                                  // vector of the Riccati recursion
  float qa[PD + 1][4], qq[PD + 1], kq[KD + 1];
  V4<float> lq[KD + 1];
  double gf = 0, vm = 0;
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int id = N - 1 - d > 0 ? N - 1 - d : 0;
    km.load(A, id, qa[d]);
    qq[d] = A[L.qv + id * VS + c];
  }
#pragma unroll
  for (int d = 0; d < KD; ++d) {
    const int id = N - 1 - d > 0 ? N - 1 - d : 0;
    kq[d] = Kb[L.K + id * KS + h * ABW + c];
    lq[d] = ld4(Kb, L.Linv + id * 16 + lj * 4);
  }
#pragma unroll MPCQ_UNROLL_SWEEP
  for (int i = N - 1; i >= 0; --i) {
    const int ip = i - PD > 0 ? i - PD : 0, ik = i - KD > 0 ? i - KD : 0;
    km.load(A, ip, qa[PD]);
    qq[PD] = A[L.qv + ip * VS + c];
    kq[KD] = Kb[L.K + ik * KS + h * ABW + c];
    lq[KD] = ld4(Kb, L.Linv + ik * 16 + lj * 4);
    const double dxc = D[L.dxd + i * VS + c];
    const double gvc = ucol ? ru * D[L.zd + i * NU + j] + (double)S[L.r0 + i * NU + j] : 0.0;
    const float a = S[L.act + i * NU + j], rtj = S[L.rt + i * NU + lj];
    // ---- gradient (double)
    double pi[4];
    l2g<float>(pc, h, pi);
    const double t = hsum(((double)qa[0][0] * pi[0] + (double)qa[0][1] * pi[1]) + ((double)qa[0][2] * pi[2] + (double)qa[0][3] * pi[3]));   // (AB''^T p)[c]
    const double g = t + gvc;
    if (lane < VS) S[L.grad + i * VS + lane] = (float)g;
    if (ucol) {
      double ag = fabs(g);
      if (!(ag == ag)) ag = 1e308;
      if (a == 0.0f) gf = ag > gf ? ag : gf;
      else { const double v = a < 0.0f ? -g : g; vm = v > vm ? v : vm; gf = ag >= 1e308 ? ag : gf; }
    }
    pc = (arow ? t : (prow ? pc : 0.0)) + (qdc * dxc + (double)qq[0]);
    // ---- feed-forward of the correction (float): rho_i = the gradient just formed
    float pv[4];
    l2g<float>(pr, h, pv);
    const float tr = hsum((qa[0][0] * pv[0] + qa[0][1] * pv[1]) + (qa[0][2] * pv[2] + qa[0][3] * pv[3]));
    const float gt = tr + (ucol ? (float)g : 0.0f);              // gt_j = rho_j + (B^T p)_j on lane column 10 + j
    float r = dpp_rows<0x12F, 0xA>(gt, h);
    r = dpp_rows<0x12E, 0xC>(r, h);
    const float gh = dpp<0x15A>(r);                              // row h: gt_h on every lane
    pr = (arow ? tr : (prow ? pr : 0.0f)) + hsum(kq[0] * gh);    // p_i = A^T p_{i+1} + K^T gt (pinned rows of K are 0)
    float gg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gg[q] = bc(gt, 10 + q);
    if (lane < NU) {
      const V4<float> li = lq[0];
      const float kvj = -(li.a * gg[0] + li.b * gg[1] + li.c * gg[2] + li.d * gg[3]);
      S[L.vin + i * VS + lane] = rtj < 0.0f ? 0.0f : kvj;
    }
    shift<float, PD>(qa);
#pragma unroll
    for (int d = 0; d < PD; ++d) qq[d] = qq[d + 1];
#pragma unroll
    for (int d = 0; d < KD; ++d) { kq[d] = kq[d + 1]; lq[d] = lq[d + 1]; }
  }
  __syncthreads();
  gF = wave_max(gf);
  vmax = wave_max(vm);
}

// Active-set method of the mixed-precision mode (TQ = float).  Working-set logic as in the fp64 method's ancestors: affine first pass of a
// warm attempt, Newton steps of the float factorisation with a bound test, pins, releases of wrong-signed multipliers at a minimiser --
// with the iterate in double (D[L.zd], D[L.dxd]) and every residual evaluated in double.  A pass at a minimiser candidate (no pin in the
// step that led to it) is one step of iterative refinement: gradient + feed-forward in one backward sweep (adjoint64_bwd), correction and
// trajectory update in one forward sweep (riccati_forward64).  It settles when the correction it has just taken is below tol_c (what is
// left is that times the contraction, a few per cent), or when the residual says the previous one already was.  Multiplier signs are
// judged on the double values.  A warm solve without a change of the working set costs one factorisation and three sweeps.
// On success D[L.zd], D[L.dxd] hold the solution, S[L.z] its float image.
// Mixed precision, a working set that cycles at the most careful release level (the sign test has been loosened once already): the classical
// primal rule from there on -- a Newton step that leaves the box is taken up to the FIRST bound it meets and only that input is pinned
// (instead of clipping and pinning every violator at once, which is what makes the float method cycle when its Newton steps are too
// inaccurate to tell which of two nearly tied inputs blocks first).  0: off (round 5 behaviour).
#ifndef MPCQ_MIXED_ONEPIN
#define MPCQ_MIXED_ONEPIN 1
#endif
#ifndef MPCQ_MIXED_TOLC
#define MPCQ_MIXED_TOLC 1e-6   // (1e-5 until round 6: see the weak-multiplier note in polish_mixed; 1e-6 costs nothing measurable)
#endif
template <typename C, bool GAB = C::GAB>
MPCQ_PHASE bool polish_mixed(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, P<float> Kb, const Lds& L, float gm, int& passes, const bool warm,
                             const int max_passes, int& why, int& converged, const bool last_resort PF_ARG) {
  using TQ = float;
  why = QPX_BUDGET;
  const int N = cN<C>(m), nv = N * NU, tid = lane_id();
  auto lbd = [&](int i) { return m.ulb[i & 3] - D[L.U + i]; };   // bounds of du in double (S[L.lb], S[L.ub] are their float images)
  auto ubd = [&](int i) { return m.uub[i & 3] - D[L.U + i]; };
  if (warm) {   // working set = inputs the previous iterate left exactly on a bound; start from z = 0 (feasible)
    for (int i = tid; i < nv; i += 64) S[L.act + i] = S[L.lb + i] == TQ(0) ? TQ(-1) : (S[L.ub + i] == TQ(0) ? TQ(1) : TQ(0));
  } else {      // working set identified by the interior point, which is also the starting point
    for (int i = tid; i < nv; i += 64) {
      S[L.act + i] = S[L.ll + i] > m.pin_ratio * S[L.sl + i] ? TQ(-1) : (S[L.lu + i] > m.pin_ratio * S[L.su + i] ? TQ(1) : TQ(0));
      D[L.zd + i] = (double)S[L.z + i];
    }
  }
  double tolm = 1e-12 * (double)gm;   // multiplier sign test: the multipliers come from double residuals (the fp64 method: 64 eps gm = 7e-15 gm)
  // The anti-cycling rule below loosens the sign test by factors of 100.  It was made for a degenerate input (a multiplier of zero to
  // rounding: released, its Newton step leaves it on its bound to rounding, pinned again, ...), where any working set of the cycle is the
  // answer.  It also fires when the working set cycles because the float factorisation is no contraction any more (predictions that
  // tumble, gradient scale above 1e6) -- and what the method then settles on can hold a wrong-signed multiplier of any size (the reference's
  // own traj2_v10_a10_gp2 flight, steps 116 and 120: 0.13 / 0.43 of full thrust off, status 0 until round 6).  So a solve that settles
  // under a loosened test says what the multipliers it ignores are worth: releasing pinned input a alone moves it by lambda_a over its
  // curvature R_aa + (B'PB)_aa (S[L.curv], left by the factorisation of this working set); beyond tol_c the solve is reported
  // (MPCQ_SOLVE_LOW_ACCURACY), never returned as a clean one.
  int loosened = 0;
  // bound proximity of a pin and what a refinement may leave behind, in units of full thrust: the iterate is double, so neither needs the
  // float-sized 1e-6 of the first version
  const double tolb = 1e-9;
  const double tol_c = MPCQ_MIXED_TOLC, tol_z = 1e-7;
  bool refactor = true, settled = false, full = false, released = false;
  bool need_roll = true;              // D[L.dxd] is not the state trajectory of D[L.zd] (z was changed outside a sweep)
  int nact = 1, careful = 0;
  double gF_prev = -1, dz_prev = 0;   // residual and correction of the previous refinement step on this working set (< 0: none)
  auto set_dx0 = [&]() { if (tid < VS) D[L.dxd + tid] = tid < NX ? D[L.x0 + i2o(tid)] - D[L.X + i2o(tid)] : 0.0; };   // dx_0 = x_meas - X_0, in double
  // Release of wrong-signed multipliers at a minimiser of the working set, by level of caution:
  //   0  every wrong-signed one (the fast path: one pass settles what the other levels take several for);
  //   1  per rotor only the worst one (the rule of the fp64 method: the multipliers of one rotor's run of saturated stages are strongly
  //      coupled; input i belongs to rotor i & 3 = lane & 3), and every WEAK one with them (under 1e-4 of the gradient scale);
  //   2  per rotor the worst one, only rotors whose worst is within 1.6x of the overall worst;
  //   3  the worst one overall (the classical rule).
  // The level rises when a release bounces (the freed inputs get pinned again) and when its Newton step is wild (below).  Thresholds and
  // candidates both come from the float images of the multipliers in S[L.grad], so the worst one always passes its own threshold.
  int rel_idx = -1;
  auto release = [&](const int level) {
    float vr = 0;
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
      if (a != TQ(0)) vr = tmax(vr, a < 0 ? -g : g);
    }
    const float vall = wave_max(vr);
    float rel_thr = level == 0 ? 0.0f : vall;
    if (level == 1 || level == 2) {
      const float w0 = wave_max((tid & 3) == 0 ? vr : 0.0f), w1 = wave_max((tid & 3) == 1 ? vr : 0.0f),
                  w2 = wave_max((tid & 3) == 2 ? vr : 0.0f), w3 = wave_max((tid & 3) == 3 ? vr : 0.0f);
      rel_thr = (tid & 3) == 0 ? w0 : ((tid & 3) == 1 ? w1 : ((tid & 3) == 2 ? w2 : w3));
      if (level == 2) rel_thr = tmax(rel_thr, 0.625f * vall);
    }
    const float weak = level == 1 ? 1e-4f * gm : 0.0f;
    int ri = -1;
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
      const float v = a < 0 ? -g : g;
      if (a != TQ(0) && (double)v > tolm && (v >= rel_thr || v <= weak)) { S[L.act + i] = 0; ri = i; }
    }
    rel_idx = level == 3 ? wave_max(ri) : -1;   // (level 3 releases one input: the one a single-pin step may meet again)
  };
  // max_passes budgets FACTORISATIONS (as in the fp64 method, where a pass is one): a refinement step reuses the factorisation at hand and is
  // not charged -- a warm solve without a change of the working set is two trips of this loop and one factorisation.  (Charging trips made
  // every solve behind a fallback fail its one-pass retry and fall back again, period after period: 8 % of the quadrotor-steps.)
  int nfac = 0;
  const int trip_cap = 3 * max_passes + 4;
  for (passes = 0; passes < trip_cap; ++passes) {
    const bool aff = warm && passes == 0;
    bool corrected = false;
    double gF = 0, vmax = 0;
    if (aff) {
      nfac += 1;
      // Warm start from z = 0 with every pinned input at a bound of exactly 0: the minimiser on the working set is the solution of the
      // affine LQ problem itself (gaps c_i, gradients q_i, r_i in the recursion): one factorisation and one forward sweep.
      int na = 0;
      for (int i = tid; i < nv; i += 64) { S[L.rho + i] = S[L.r0 + i]; na += S[L.act + i] != TQ(0) ? 1 : 0; }
      nact = wave_sum(na);
      for (int it = tid; it < N * VS; it += 64) S[L.Dx + it] = (it & 15) < NX ? A[L.c + it] : TQ(0);   // gap operand of the factorisation
      __syncthreads();
      PF_START();
      TQ gfac = 0;
      const bool fok = riccati_factor<C, true, true>(m, S, A, Kb, L PF_PASS, &gfac);
      PF_STOP(PF_FACTOR);
      if (!fok) { why = QPX_NUMERIC; return false; }
      gm = tmax(TQ(1), gfac);
      tolm = 1e-12 * (double)gm;
      refactor = false;
      set_dx0();   // (behind the factorisation: D[L.dxd] shares its space with the gap operand)
      __syncthreads();
      PF_START(); riccati_forward64<C, true>(m, D, S, A, Kb, L PF_PASS); PF_STOP(PF_FWD);
      need_roll = false;
    } else {
      int na = 0;
      for (int i = tid; i < nv; i += 64) {
        const TQ a = S[L.act + i];
        if (a < 0) D[L.zd + i] = lbd(i);
        else if (a > 0) D[L.zd + i] = ubd(i);
        na += a != TQ(0) ? 1 : 0;
      }
      nact = wave_sum(na);
      if (need_roll) {
        set_dx0();
        __syncthreads();
        PF_START(); rollout64<C>(m, D, S, A, L); PF_STOP(PF_ROLL);
        need_roll = false;
      } else __syncthreads();
      // residual of the QP at zd in double; with a factorisation of this working set at hand, the feed-forward of the correction in the same sweep
      const bool fused = !refactor;
      PF_START();
      if (fused) adjoint64_bwd<C>(m, D, S, A, Kb, L, gF, vmax);
      else adjoint64<C>(m, D, S, A, L, gF, vmax);
      PF_STOP(PF_ADJ);
      if (!(gF < 1e300)) { why = QPX_NUMERIC; return false; }
#ifdef MPCQ_EMU_DEBUG
      if (tid == 0) printf("  mixed pass %d warm %d full %d gF %.3e vmax %.3e tolm %.3e nact %d (prev gF %.3e dz %.3e)\n", passes, (int)warm, (int)full, gF, vmax, tolm, nact, gF_prev, dz_prev);
#endif
      bool rel_now = false;
      if (full) {   // the point minimises the QP on the working set up to the error under refinement: multipliers are meaningful
        if (vmax > tolm) {
          for (int i = tid; i < nv; i += 64) S[L.z + i] = S[L.act + i];   // the working set in front of the release (S[L.z] is free while zd is the iterate)
          release(careful);
          refactor = true;
          rel_now = true;
          __syncthreads();
        } else if (gF == 0.0 || (gF_prev > 0 && dz_prev * (gF / gF_prev) <= tol_z)) {
          settled = true;   // the previous correction left less than tol_z (its size times the contraction the residual shows)
          break;
        } else if (gF_prev > 0 && gF > 0.5 * gF_prev) {
          converged = 0;    // refinement stagnates (never observed: cond x eps32 < 1): accepted and reported
          settled = true;
          break;
        }
      }
      released = rel_now;
      if (rel_now) gF_prev = -1;
      if (refactor) {
        if (nfac >= max_passes) break;   // out of budget (why = QPX_BUDGET)
        nfac += 1;
        for (int i = tid; i < nv; i += 64) S[L.rho + i] = S[L.grad + GI(i)];
        __syncthreads();
        PF_START();
        const bool fok = riccati_factor<C, true>(m, S, A, Kb, L PF_PASS);
        PF_STOP(PF_FACTOR);
        if (!fok) { why = QPX_NUMERIC; return false; }
        refactor = false;
      }
      PF_START(); riccati_forward64<C, false>(m, D, S, A, Kb, L PF_PASS); PF_STOP(PF_FWD);
      corrected = full && !rel_now;
      // A release whose Newton step leaves the box by more than half its width is a cascade in the making (the step gets clipped, the
      // violators pinned wholesale, the next multipliers are worse; seen at N = 50: 44 inputs pinned in one pass, the working set driven
      // to 199 of 200): taken back -- the next trip of the loop re-evaluates the same point and releases one level more cautiously.
      if (rel_now && careful < 3) {
        float dmx = 0;
        for (int i = tid; i < nv; i += 64)
          if (S[L.act + i] == TQ(0)) dmx = tmax(dmx, tabs(S[L.dz + i]));
        if (wave_max(dmx) > 0.5f) {
          for (int i = tid; i < nv; i += 64) S[L.act + i] = S[L.z + i];
          careful += 1;
          refactor = true; need_roll = true;   // (the sweep taken back had moved the trajectory along)
          released = false;
          __syncthreads();
          continue;
        }
      }
    }
    // full Newton step; if it leaves the box, clip and pin EVERY violator at once
    int viol = 0;
    double dzm = 0;
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const double d = aff ? D[L.zd + i] : (double)S[L.dz + i], zn = aff ? d : D[L.zd + i] + d;
      if (!(zn == zn)) viol |= 2;
      if (zn < lbd(i) || zn > ubd(i)) viol |= 1;
      dzm = fabs(d) > dzm ? fabs(d) : dzm;
    }
    viol = wave_reduce(viol, [](int a, int b) { return a | b; });
    if (viol & 2) { why = QPX_NUMERIC; return false; }
    dzm = wave_max(dzm);
    int nblk = 0, pin_idx = -1;
    double alpha_pin = 1.0;
    const bool onepin = MPCQ_MIXED_ONEPIN != 0 && !aff && careful >= 3 && loosened > 0 && viol != 0;
    if (onepin) {   // (see MPCQ_MIXED_ONEPIN) up to the first bound the step meets; that input alone joins the working set
      double amin = 2.0;
      int imin = nv;
      for (int i = tid; i < nv; i += 64) {
        if (S[L.act + i] != TQ(0)) continue;
        const double z = D[L.zd + i], d = (double)S[L.dz + i], zn = z + d, lb = lbd(i), ub = ubd(i);
        double a = 2.0;
        if (zn < lb) a = (lb - z) / d; else if (zn > ub) a = (ub - z) / d;
        a = a < 0.0 ? 0.0 : a;
        if (a < amin) { amin = a; imin = i; }
      }
      alpha_pin = wave_min(amin);
      pin_idx = wave_min(amin == alpha_pin ? imin : nv);
      for (int i = tid; i < nv; i += 64) {
        if (S[L.act + i] != TQ(0)) continue;
        const double lb = lbd(i), ub = ubd(i), d = (double)S[L.dz + i];
        double z = D[L.zd + i] + alpha_pin * d;
        z = z < lb ? lb : (z > ub ? ub : z);
        if (i == pin_idx) { z = d < 0.0 ? lb : ub; S[L.act + i] = d < 0.0 ? TQ(-1) : TQ(1); }
        D[L.zd + i] = z;
      }
      nblk = 1;
    } else {
    for (int i = tid; i < nv; i += 64) {
      if (S[L.act + i] != TQ(0)) continue;
      const double lb = lbd(i), ub = ubd(i);
      double z = aff ? D[L.zd + i] : D[L.zd + i] + (double)S[L.dz + i];
      if (viol) {
        if (z <= lb + tolb) { z = lb; S[L.act + i] = -1; nblk += 1; }
        else if (z >= ub - tolb) { z = ub; S[L.act + i] = 1; nblk += 1; }
      }
      D[L.zd + i] = z;
    }
    nblk = wave_sum(nblk);
    }
#ifdef MPCQ_EMU_DEBUG
    if (tid == 0) printf("     step %.3e nblk %d corrected %d\n", dzm, nblk, (int)corrected);
#endif
    full = nblk == 0;
    // wholesale bounce in a saturated regime: leave it to the interior point (not behind its final iterations: last_resort)
    if (released && nblk >= 8 && 2 * (nact + nblk) >= nv && !last_resort) { why = QPX_BOUNCE; return false; }
    if (nblk > 0) {   // clipped inputs: the sweep's trajectory is not theirs
      refactor = true; need_roll = true; gF_prev = -1;
      if (released) {
        // a release that bounces straight back raises the level of caution; at the last level -- the single worst multiplier released, and
        // pinned again by a step that leaves the box by a rounding error -- the input is degenerate (it sits on its bound with a multiplier
        // of zero to rounding): the sign test is loosened until the pair stops trading places (seen on the bench workload's own seed: one
        // input released and re-pinned sixty times until the budget was gone)
        // (single-pin steps: a release followed by a pin is the method at work, not a bounce -- unless the step pins the input just
        //  released without having moved: that one is degenerate)
        if (careful < 3) careful += 1;
        else if (!onepin || (pin_idx == rel_idx && alpha_pin < 1e-9)) { tolm *= 100.0; loosened += 1; }
      }
    }
    released = false;
    if (corrected && full) {
      if (dzm <= tol_c) {   // (the forward sweep has taken the trajectory along)
        // The multipliers behind this were evaluated in FRONT of the correction, at a point off by dzm: that moves the multiplier of pinned input a
        // by about its curvature R_aa + (B'PB)_aa times dzm (S[L.curv]), enough to hide the wrong sign of a weak one -- an input that belongs a few
        // 1e-5 inside its bound stays pinned, the solve settles 1e-5 off with status 0 (round 6, every solve of the bench workload against the
        // fp64 engine: 6 in 1.7 M, idling quadrotors, tools/f32_audit.py).  A solve with such a multiplier takes one more trip: the residual
        // at the refined point shows its sign.
        int weak = 0;
        if (nact > 0)
          for (int i = tid; i < nv; i += 64) {
            const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
            if (a != TQ(0) && (a < 0 ? -g : g) > -8.0f * (S[L.curv + i] + S[L.wq + 2 * VS + (i & 3)]) * (float)dzm) weak = 1;
          }
        if (nact > 0) weak = wave_max(weak);
        if (!weak) { settled = true; passes += 1; break; }
      }
      gF_prev = gF; dz_prev = dzm;
    }
    __syncthreads();
  }
  __syncthreads();
  if (settled && loosened > 0) {   // (see `loosened` above)
    float est = 0;
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i], g = S[L.grad + GI(i)];
      const float v = a < 0 ? -g : g;
      if (a != TQ(0) && v > 0.0f) est = tmax(est, v / (S[L.curv + i] + S[L.wq + 2 * VS + (i & 3)]));
    }
    est = wave_max(est);
    // (handing such a solve to the next stage instead -- interior point, then this method again -- was tried: it rescued none and turned
    //  solves that were within the budget into MPCQ_SOLVE_MAXITER)
    if (!(est <= (float)tol_c)) converged = 0;
  }
  if (settled)
    for (int i = tid; i < nv; i += 64) {
      const TQ a = S[L.act + i];
      S[L.z + i] = a < 0 ? S[L.lb + i] : (a > 0 ? S[L.ub + i] : (TQ)D[L.zd + i]);
    }
  __syncthreads();
  return settled;
}

// ------------------------------------------------------------------ fp64 instances: the interior point of the fallback in float
// What the active-set method behind the interior point takes over is a WORKING SET (which inputs sit on which bound), checked and
// corrected by its own double factorisations -- the answer of a fallback solve never comes from the interior point.  Its iterations
// therefore run in float on the matrix cores' float tiles (v_mfma_f32_16x16x4_f32: half the passes of the f64 tile, 32-bit DPP and
// LDS traffic), with the operands read from the double stage records and rounded in registers.  The float vectors live in the space of
// the double ones (float array X = the first half of the bytes of double array X: offsets x 2 in a float view of the same LDS), so the
// layout and the LDS budget of the instance do not change.  The interior point is the one the float instances run (ipm_run_regs<float>:
// centring floor, float residual test).  Returns false when it broke down (the caller then runs the double interior point from its
// start): the double start vectors are overwritten either way.  -DMPCQ_HYBRID_IPM=0: the double interior point (rounds 1-4).
#ifndef MPCQ_HYBRID_IPM
#define MPCQ_HYBRID_IPM 1
#endif
#ifndef MPCQ_HYBRID_TOL   // hand-over tolerance of the float interior point (the float instances: 1e-5)
#define MPCQ_HYBRID_TOL 3e-7f
#endif
#ifndef MPCQ_HYBRID_RD    // what it asks of its float dual residual, in units of the gradient scale (the float instances: 3e-4)
#define MPCQ_HYBRID_RD 3e-4f
#endif
#ifndef MPCQ_HYBRID_MAXIT // its iteration cap (it needs 3 .. 10; one that stalls -- 1 fallback solve in 2e4 on the bench workload, the double one
#define MPCQ_HYBRID_MAXIT 24   // stalls on the same QPs -- hands over what it has: the active-set method verifies what it ends on itself)
#endif
template <typename C, bool GAB = C::GAB>
MPCQ_COLD int ipm_float_stage(const DevModel<double>& m, P<double> S, P<double> A, P<double> Kb, const Lds& L, const double gm, int& it PF_ARG) {
  constexpr int N = C::N, nv = N * NU, R = (nv + 63) / 64, RG = (N * VS + 63) / 64;
  const int tid = lane_id();
  Lds Lf = L;   // stage records: the double ones (offsets into A unchanged); QP workspace and gains: the float view
  Lf.z = 2 * L.z; Lf.sl = 2 * L.sl; Lf.su = 2 * L.su; Lf.ll = 2 * L.ll; Lf.lu = 2 * L.lu; Lf.dza = 2 * L.dza; Lf.dz = 2 * L.dz;
  Lf.rho = 2 * L.rho; Lf.act = 2 * L.act; Lf.rt = 2 * L.rt; Lf.grad = 2 * L.grad; Lf.vin = 2 * L.vin; Lf.dx = 2 * L.dx; Lf.Dx = 2 * L.Dx;
  Lf.K = 2 * L.K; Lf.Linv = 2 * L.Linv; Lf.sF = 2 * L.sF; Lf.sT = 2 * L.sT; Lf.stv = 2 * L.stv; Lf.wq = 2 * L.wq;
  const P<float> Sf = as_float(S), Kf = as_float(Kb);
  // the start, read in double and rounded behind a barrier (a float array overlaps the double elements of other lanes)
  double zr[R], slr[R], sur[R], llr[R], lur[R], gr[RG];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + 64 * r, ii = i < nv ? i : 0;
    zr[r] = S[L.z + ii]; slr[r] = S[L.sl + ii]; sur[r] = S[L.su + ii]; llr[r] = S[L.ll + ii]; lur[r] = S[L.lu + ii];
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) { const int k = tid + 64 * r; gr[r] = S[L.grad + (k < N * VS ? k : 0)]; }
  const double wqr = S[L.wq + (tid < 3 * VS ? tid : 0)];   // live in double behind the interior point: kept here, put back below
  const double dx0r = S[L.dx + (tid < VS ? tid : 0)];       // dx_0 likewise
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + 64 * r;
    if (i < nv) { Sf[Lf.z + i] = (float)zr[r]; Sf[Lf.sl + i] = (float)slr[r]; Sf[Lf.su + i] = (float)sur[r]; Sf[Lf.ll + i] = (float)llr[r]; Sf[Lf.lu + i] = (float)lur[r]; }
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) { const int k = tid + 64 * r; if (k < N * VS) Sf[Lf.grad + k] = (float)gr[r]; }
  if (tid < 3 * VS) Sf[Lf.wq + tid] = (float)wqr;
  for (int i = tid; i < (N + 1) * VS; i += 64) Sf[Lf.dx + i] = 0;   // (the float state trajectory is bookkeeping nobody reads)
  __syncthreads();
  int itf = 0;
  const int stf = ipm_run_regs<C, float, GAB>(m, Sf, A, Kf, Lf, MPCQ_HYBRID_TOL, (float)gm, itf PF_PASS, MPCQ_HYBRID_RD, MPCQ_HYBRID_MAXIT);
  it += itf;
  // back to double: the smaller slack of an input is taken as it is, the point and the other slack follow from it (both positive, and
  // consistent with the bounds to double rounding -- what a continuation in double needs)
  float zf[R], slf[R], suf[R], llf[R], luf[R];
  int sound = 1;   // every slack and multiplier finite and positive (an iteration that ended at its cap is handed over only then)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + 64 * r, ii = i < nv ? i : 0;
    zf[r] = Sf[Lf.z + ii]; slf[r] = Sf[Lf.sl + ii]; suf[r] = Sf[Lf.su + ii]; llf[r] = Sf[Lf.ll + ii]; luf[r] = Sf[Lf.lu + ii];
    if (!(slf[r] > 0.0f && slf[r] < 1e30f && suf[r] > 0.0f && suf[r] < 1e30f && llf[r] > 0.0f && llf[r] < 1e30f && luf[r] > 0.0f && luf[r] < 1e30f)) sound = 0;
  }
  sound = wave_min(sound);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + 64 * r;
    if (i < nv) {
      const double lb = S[L.lb + i], ub = S[L.ub + i];
      double z, sl, su;
      if (slf[r] <= suf[r]) { sl = (double)slf[r]; z = lb + sl; su = ub - z; }
      else { su = (double)suf[r]; z = ub - su; sl = z - lb; }
      S[L.z + i] = z; S[L.sl + i] = sl; S[L.su + i] = su; S[L.ll + i] = (double)llf[r]; S[L.lu + i] = (double)luf[r];
    }
  }
  if (tid < 3 * VS) S[L.wq + tid] = wqr;
  if (tid < VS) S[L.dx + tid] = dx0r;
  __syncthreads();
  return (stf == 2 && !sound) ? 1 : stf;   // 0: at its tolerance | 2: at its iteration cap with a sound iterate | else: broken down
}

// Box-QP solve.  (1) Warm active-set attempt: the RTI iterate is persisted, so the working set of the
// previous control step (inputs sitting exactly on a bound) is usually still optimal or off by one or two
// inputs; a few passes of the active-set method from z = 0 then end on the exact KKT point at the cost of
// about one factorisation.  (2) Otherwise (cold start, large changes, degenerate cycling): Mehrotra IPM to
// the hand-over tolerance, then the active-set polish from the IPM's working set; if that does not settle
// either, IPM iterations down to the final tolerance.  The QP is strictly convex, so every branch ends on
// the same unique optimum.  On exit S[L.z] holds the solution and S[L.dx] the matching state trajectory;
// returns passes (+1000 when the warm attempt had to fall back).
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
MPCQ_PHASE int solve_qp(const DevModel<TQ>& m, P<double> D, P<TQ> S, P<TQ> A, P<TQ> G, const Lds& L, int* status, const int prev_iter, int* work PF_ARG) {
  const int N = cN<C>(m), nv = N * NU, tid = lane_id();
  int it = 0, passes = 0, wpasses = 0, why = 0;
  TQ gm = 1;
  const P<TQ> Kb = C::GK ? G : S;   // where the gains live
  static_assert(sizeof(TQ) == 8, "float instances solve their QP in solve_qp_mixed");
  // prev_iter: this quadrotor's previous return value (0: cold start).  Decimal fields (qp_iter of include/mpcq.h):
  //   passes + interior-point iterations | x 1000: the warm attempt was given up or skipped (fallback solve) |
  //   x 10000: flip mark | x 100000: why the warm attempt ended (QPX_*).
  // A quadrotor whose references are out of reach (inputs saturated over most of the horizon, the working set changing by
  // many inputs every period) fails the warm attempt period after period: after a fallback the next attempt is short
  // (warm_retry passes), so that such a quadrotor costs its launch one interior-point solve, not that plus a long
  // active-set attempt; the first period in which the short attempt succeeds restores the full budget.
  // Flip mark: the previous solve fell back AND its solution differed from the one before in more than flip_max
  // bound states -- a quadrotor whose saturated inputs flip between rotors from period to period (infeasible references,
  // near-degenerate QPs): the previous working set is no guess at all there, the warm attempt is skipped altogether.
  const bool flipping = m.flip_max >= 0 && (prev_iter / 10000) % 10 != 0;
  const int warm_cap = flipping ? 0 : ((prev_iter / 1000) % 10 != 0 ? m.warm_retry : m.warm_max);
  if (flipping) { wpasses = 1000; why = QPX_SKIPPED; }   // counts as a fallback solve
  if (prev_iter > 0 && warm_cap > 0) {
    if (polish<C>(m, S, A, G, L, gm, wpasses, true, warm_cap, why PF_PASS)) {   // sets z = 0 and its own gradient scale
      *status = 0;
      *work = wpasses | (wpasses << 16);   // one (possibly resumed) factorisation and one forward sweep per pass
      return wpasses;
    }
    wpasses += 1000;
  }
  int st = 0;
  // interior start
  auto interior_start = [&]() {
    for (int i = tid; i < nv; i += 64) {
      const TQ lb = S[L.lb + i], ub = S[L.ub + i], w = ub - lb;
      const TQ z0 = tmin(tmax(TQ(0), lb + m.ipm_margin * w), ub - m.ipm_margin * w);
      S[L.z + i] = z0; S[L.sl + i] = z0 - lb; S[L.su + i] = ub - z0;
    }
    __syncthreads();
    PF_START(); rollout<C>(m, S, A, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
    PF_START(); adjoint<C>(m, S, A, L); PF_STOP(PF_ADJ);
    gm = 1;
    for (int i = tid; i < nv; i += 64) gm = tmax(gm, tabs(S[L.grad + GI(i)]));
    gm = wave_max(gm);
    for (int i = tid; i < nv; i += 64) { S[L.ll + i] = m.ipm_mu0 * gm / S[L.sl + i]; S[L.lu + i] = m.ipm_mu0 * gm / S[L.su + i]; }
    __syncthreads();
  };
  interior_start();
  DBG_DUMP(1, 0, S, L.z, nv); DBG_DUMP(1, 128, S, L.sl, nv); DBG_DUMP(1, 256, S, L.su, nv); DBG_DUMP(1, 384, S, L.ll, nv); DBG_DUMP(1, 512, S, L.lu, nv);
  DBG_DUMP(1, 640, S, L.grad, N * VS); DBG_DUMP(1, 1024, S, L.dx, (N + 1) * VS); DBG_DUMP(1, 1536, A, L.AB, N * ABS > 2500 ? 2500 : N * ABS);
  // shapes whose interior point runs in registers: its iterations to the hand-over in float (ipm_float_stage); broken down -> in double from the start
  bool handed = false, broke = false;
  int fit = 0;   // interior-point iterations executed in float (reported in the work word: what a latency model prices with the float chains)
  if constexpr (MPCQ_HYBRID_IPM != 0 && C::N > 0 && C::N * NU <= 128) {
    if (m.polish_max > 0) {
      const int stf = ipm_float_stage<C>(m, S, A, Kb, L, gm, it PF_PASS);
      fit = it;                        // (the interior point's iterations so far are the float ones)
      handed = stf == 0 || stf == 2;   // (2: stalled short of its tolerance at its iteration cap -- the working set it indicates is tried all the same)
      if (!handed) { broke = true; interior_start(); }
    }
  }
  if (!handed) {   // (behind a float interior point that broke down -- possibly at its iteration cap -- the double one has its own budget)
    int itd = broke ? 0 : it;
    st = ipm_run<C>(m, S, A, Kb, L, m.polish_max > 0 ? m.ipm_tol : m.qp_tol, gm, itd PF_PASS);
    it = broke ? it + itd : itd;
  }
  bool need_roll = true;
  // (st == 2: the double interior point at its iteration cap -- it stalls on about one fallback solve in 2e4 of the bench workload, the dual
  //  residual lingering above its tolerance with the complementarity long there; the iterate is feasible and indicates a working set, and
  //  what the active-set method settles on it has verified itself)
  if ((st == 0 || st == 2) && m.polish_max > 0) {
    for (int i = tid; i < nv; i += 64) S[L.dza + i] = S[L.z + i];
    __syncthreads();
    int why2 = 0;
    if (polish<C>(m, S, A, G, L, gm, passes, false, m.polish_max, why2 PF_PASS)) { need_roll = false; st = 0; }
    else {
      for (int i = tid; i < nv; i += 64) S[L.z + i] = S[L.dza + i];
      __syncthreads();
      PF_START(); rollout<C>(m, S, A, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
      PF_START(); adjoint<C>(m, S, A, L); PF_STOP(PF_ADJ);
      int itc = handed ? 0 : it;   // (the iterations of a float interior point are not charged to the double one that continues from its iterate)
      st = ipm_run<C>(m, S, A, Kb, L, m.qp_tol, gm, itc PF_PASS);
      it = handed ? it + itc : itc;
    }
  }
  if (need_roll) { PF_START(); rollout<C>(m, S, A, L, L.dx, L.z, true); PF_STOP(PF_ROLL); }   // state trajectory of the returned z
  *status = st;
  // bound states that differ between the previous solution (z = 0: on a bound where lb or ub is 0) and this one
  int chg = 0;
  if (m.flip_max >= 0 && prev_iter > 0) {
    for (int i = tid; i < nv; i += 64) {
      const TQ lb = S[L.lb + i], ub = S[L.ub + i], z = S[L.z + i];
      const int was = lb == TQ(0) ? -1 : (ub == TQ(0) ? 1 : 0), is = z == lb ? -1 : (z == ub ? 1 : 0);
      chg += was != is ? 1 : 0;
    }
    chg = wave_sum(chg);
  }
  {   // factorisations: warm passes + interior-point iterations + passes behind it; sweeps: one per pass, two forward + one backward
      // (+ the adjoint in fp32) per interior-point iteration, rollout + adjoint of the interior start, the final rollout
    const int wp = wpasses % 1000;
    // (bit 15: the float interior point broke down and the double one ran from the start -- its iterations are counted too)
    const unsigned sweeps = (unsigned)(wp + passes + 3 * it + 2 + (need_roll ? 1 : 0) + (broke ? 2 : 0));
    *work = (int)((unsigned)(it + passes + wp) | (broke ? 0x8000u : 0u) | ((sweeps < 2047u ? sweeps : 2047u) << 16) | ((unsigned)(fit < 31 ? fit : 31) << 27));
  }
  return it + passes + wpasses + (m.flip_max >= 0 && chg > m.flip_max ? 10000 : 0) + 100000 * why;
}

// Box-QP solve of the float instances (mixed precision).  Same three stages as solve_qp -- (0) warm active-set attempt, (1) interior
// point to the hand-over tolerance + active set from its working set, (2) interior point to its last tolerance + active set without the
// early exit -- written as ONE loop so that the interior point and the active-set method are each instantiated once in the kernel (as four
// inlined copies of polish_mixed the float instances were 370 KB of code and lost 8 % to instruction fetch).  What differs from fp64: the
// answer always comes from the active-set method (its iterate and residuals are double); an interior point that breaks down in float --
// a stage Hessian losing definiteness, the iteration cap -- leaves a feasible iterate from which stage 2 takes over; only if the
// active-set method fails behind stage 2 as well is the interior point's own answer taken, and reported (MPCQ_SOLVE_LOW_ACCURACY).
template <typename C, bool GAB = C::GAB>
MPCQ_PHASE int solve_qp_mixed(const DevModel<float>& m, P<double> D, P<float> S, P<float> A, P<float> G, const Lds& L, int* status, const int prev_iter, int* work PF_ARG) {
  using TQ = float;
  const int N = cN<C>(m), nv = N * NU, tid = lane_id();
  int it = 0, passes = 0, wpasses = 0, why = 0, conv = 1, st = 0;
  TQ gm = 1;
  const P<TQ> Kb = C::GK ? G : S;
  const bool flipping = m.flip_max >= 0 && (prev_iter / 10000) % 10 != 0;   // (prev_iter, flip mark, budgets: see solve_qp)
  const int warm_cap = flipping ? 0 : ((prev_iter / 1000) % 10 != 0 ? m.warm_retry : m.warm_max);
  if (flipping) { wpasses = 1000; why = QPX_SKIPPED; }
  bool solved = false, ipm_sound = true;
  for (int stage = (prev_iter > 0 && warm_cap > 0) ? 0 : 1; stage <= 2 && !solved && ipm_sound; ++stage) {
    int cap = warm_cap;
    if (stage >= 1) {
      // D[L.dxd] shares its space with the float vectors dx | Dx: behind an active-set attempt dx_0 = x_meas - X_0 is put back
      if (tid < VS) S[L.dx + tid] = tid < NX ? (TQ)(D[L.x0 + i2o(tid)] - D[L.X + i2o(tid)]) : TQ(0);
      if (stage == 1) {   // interior start: the previous solution (z = 0) pushed inside the box
        for (int i = tid; i < nv; i += 64) {
          const TQ lb = S[L.lb + i], ub = S[L.ub + i], w = ub - lb;
          const TQ z0 = tmin(tmax(TQ(0), lb + m.ipm_margin * w), ub - m.ipm_margin * w);
          S[L.z + i] = z0; S[L.sl + i] = z0 - lb; S[L.su + i] = ub - z0;
        }
      } else {            // continue from the interior point's last iterate
        for (int i = tid; i < nv; i += 64) S[L.z + i] = S[L.dza + i];
      }
      __syncthreads();
      PF_START(); rollout<C>(m, S, A, L, L.dx, L.z, true); PF_STOP(PF_ROLL);
      PF_START(); adjoint<C>(m, S, A, L); PF_STOP(PF_ADJ);
      if (stage == 1) {
        gm = 1;
        for (int i = tid; i < nv; i += 64) gm = tmax(gm, tabs(S[L.grad + GI(i)]));
        gm = wave_max(gm);
        for (int i = tid; i < nv; i += 64) { S[L.ll + i] = m.ipm_mu0 * gm / S[L.sl + i]; S[L.lu + i] = m.ipm_mu0 * gm / S[L.su + i]; }
        __syncthreads();
      }
      st = ipm_run<C>(m, S, A, Kb, L, (stage == 1 && m.polish_max > 0) ? m.ipm_tol : m.qp_tol, gm, it PF_PASS);
      if (st != 0) {   // broken down before its tolerance: is the iterate it leaves usable?
        int finite = (st == 4 || st == 2) ? 1 : 0;
        for (int i = tid; i < nv; i += 64) { const TQ v = S[L.z + i]; if (!(tabs(v) < TQ(1e30)) || !(S[L.sl + i] > TQ(0)) || !(S[L.su + i] > TQ(0))) finite = 0; }
        ipm_sound = wave_min(finite) != 0;
        stage = 2;       // whatever follows is the last resort
      }
      if (!ipm_sound || m.polish_max <= 0) break;
      for (int i = tid; i < nv; i += 64) S[L.dza + i] = S[L.z + i];
      __syncthreads();
      cap = stage == 1 ? m.polish_max : 4 * m.polish_max;
    }
    int np = 0, wy = 0;
    solved = polish_mixed<C>(m, D, S, A, Kb, L, gm, np, stage == 0, cap, wy, conv, stage == 2 PF_PASS);
    if (stage == 0) { wpasses = np; if (!solved) { why = wy; wpasses += 1000; } }
    else passes += np;
  }
  if (solved) st = 0;
  else {   // the interior point's own answer (or, behind a breakdown, its last iterate): reported
    for (int i = tid; i < nv; i += 64) D[L.zd + i] = (double)S[L.z + i];
    if (tid < VS) D[L.dxd + tid] = tid < NX ? D[L.x0 + i2o(tid)] - D[L.X + i2o(tid)] : 0.0;
    __syncthreads();
    PF_START(); rollout64<C>(m, D, S, A, L); PF_STOP(PF_ROLL);
    conv = 0;
  }
  *status = (st == 0 && !conv) ? 8 : st;
  // bound states that differ between the previous solution (z = 0: on a bound where lb or ub is 0) and this one
  int chg = 0;
  if (m.flip_max >= 0 && prev_iter > 0 && wpasses >= 1000) {
    for (int i = tid; i < nv; i += 64) {
      const TQ lb = S[L.lb + i], ub = S[L.ub + i], z = S[L.z + i];
      const int was = lb == TQ(0) ? -1 : (ub == TQ(0) ? 1 : 0), is = z == lb ? -1 : (z == ub ? 1 : 0);
      chg += was != is ? 1 : 0;
    }
    chg = wave_sum(chg);
  }
  {   // (counts as in solve_qp; a refinement step is charged as a pass: two sweeps, no factorisation)
    const int wp = wpasses % 1000;
    *work = (it + passes + wp) | ((2 * (wp + passes) + 4 * it + 2) << 16);
  }
  return it + passes + wpasses + (m.flip_max >= 0 && chg > m.flip_max ? 10000 : 0) + 100000 * why;
}

// ------------------------------------------------------------------ RGP regress (3 axes, one new point each)
// RGP.regress / RGP.predict of the reference (src/gp/RGP.py:199-208,303-330), scalar new point:
//   J = k* Kx^-1 ; mu_p = J mu ; Cp = sf2 - J k* + J C J^T ; G = C J^T/(Cp + sn2) ;
//   mu += G (y - mu_p) ; C -= G (J C)      (not symmetrised, as in the reference)
template <typename C, typename TQ = typename C::T>
MPCQ_PHASE void rgp_regress(const DevModel<TQ>& m, P<TQ> S, const Lds& L, P<TQ> gmu, P<TQ> gC, P<const double> vb, P<const double> ad, bool c_staged = false) {
  const int n = cNB<C>(m), tid = lane_id(), NT = blockDim.x, n3 = 3 * n, nn = n * n;
  const P<TQ> Cw = S + L.rgp;
  const P<TQ> ks = Cw + al4(3 * nn);
  const P<TQ> Jt = ks + al4(n3);
  const P<TQ> JC = Jt + al4(n3);
  const P<TQ> CJ = JC + al4(n3);
  const P<TQ> mu = CJ + al4(n3);
  const P<TQ> sc = mu + al4(n3);
  const P<const TQ> Kxinv = mk(m.Kxinv, 3L * nn, CK_KXINV), basis = mk(m.basis, n3, CK_BASIS);
  if (!c_staged)
    for (int i = tid; i < 3 * nn; i += NT) Cw[i] = gC[i];
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n;
    const TQ dl = (TQ)vb[d] - basis[i];
    ks[i] = m.sf2[d] * texp(TQ(-0.5) * dl * dl * m.L2inv[d]);
    mu[i] = gmu[i];
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    TQ t = 0;
    for (int k = 0; k < n; ++k) t += ks[d * n + k] * Kxinv[d * nn + k * n + j];
    Jt[i] = t;
  }
  __syncthreads();
  for (int i = tid; i < n3; i += NT) {
    const int d = i / n, j = i % n;
    TQ t = 0, s = 0;
    for (int k = 0; k < n; ++k) { t += Jt[d * n + k] * Cw[d * nn + k * n + j]; s += Cw[d * nn + j * n + k] * Jt[d * n + k]; }
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
    gmu[i] = st16(mu[i] + CJ[i] * sc[d * 4 + 1] * sc[d * 4]);
  }
  for (int i = tid; i < 3 * nn; i += NT) {
    const int d = i / nn, r = (i / n) % n, c = i % n;
    gC[i] = st16(Cw[i] - CJ[d * n + r] * sc[d * 4 + 1] * JC[d * n + c]);
  }
}

// ------------------------------------------------------------------ the fused step kernel
// reference row (get_reference_chunk, src/utils/utils.py:897-931) for horizon node j
__device__ inline long chunk_row(int j, int have, int idx, int skip, int len) { return j < have ? (long)idx + (long)j * skip : (long)len - 1; }
// number of chunk rows taken from the trajectory itself (the rest repeat its last row): all N while more than N*skip
// rows are left, the clipped strided slice ceil(left/skip) while more than skip-1 are left, none after that
__device__ inline int chunk_have(int len, int idx, int N, int skip) {
  const long left = (long)len - idx;
  if (left > (long)N * skip) return N;
  if (left > skip - 1) { const long h = (left + skip - 1) / skip; return h > N ? N : (int)h; }
  return 0;
}

#ifndef MPCQ_MIN_WAVES_PER_EU
#define MPCQ_MIN_WAVES_PER_EU 1
#endif
// (compact-layout instances exist to put a second wave on a SIMD: at most 256 registers)
template <typename C, typename TQ = typename C::T, bool GAB = C::GAB>
__global__ void __launch_bounds__(64, C::GK ? 2 : MPCQ_MIN_WAVES_PER_EU) step_kernel(const DevModel<typename C::T> m, const DevState<typename C::T> st, const int mode) {
  const int tid = lane_id();
  const int N = cN<C>(m), nb = cNB<C>(m), nv = N * NU;
  const int b = st.order ? st.order[blockIdx.x] : st.b0 + (int)blockIdx.x;
  const Lds L = lds_layout(N, nb, C::LAYOUT, sizeof(TQ) == 4);
#ifdef MPCQ_CHECKED
  if (tid == 0) *reinterpret_cast<int**>(smem_raw) = st.chk;   // where violations are recorded (ck_rec)
  if (tid == 0 && b == 0 && st.chk) { const unsigned long long pc = __builtin_amdgcn_s_getpc(); st.chk[9] = (int)(unsigned)pc; st.chk[10] = (int)(unsigned)(pc >> 32); }
  __syncthreads();
#endif
  // (float instances: the double view extends over the TQ region, where the state trajectory of the refined QP solution lives: L.dxd)
  const P<double> D = mk(reinterpret_cast<double*>(smem_raw + CK_HDR), sizeof(TQ) == 4 ? (L.dbytes + L.qtotal * 4) / 8 : L.dbytes / 8, CK_LDS_D);
  const P<TQ> S = mk(reinterpret_cast<TQ*>(smem_raw + CK_HDR + L.dbytes), L.qtotal, CK_LDS_Q);
  const P<TQ> G = mk(st.stage + (size_t)b * L.gtotal, L.gtotal, CK_STAGE);   // per-instance record in global memory
  const P<TQ> A = GAB ? G : S;   // base of the stage records
  const bool gp = nb > 0;
#ifdef MPCQ_PROFILE
  Prof pf;
  for (int k = 0; k < PF_N; ++k) pf.acc[k] = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  pf.t = t_begin;
  pf.oth = 11;
#endif
  // MODE_RUN: this workgroup advances its quadrotor through run_steps control periods {step -> drag plant} on its
  // own: instances are independent, so nothing forces the batch to wait for its slowest member every period.
  const int periods = C::RUN ? st.run_steps : 1;
  for (int period = 0; period < periods; ++period) {
#ifdef MPCQ_TRACE_NAN   // reproducer builds only (tools/repro_codegen): which phase of which period first holds a non-finite value
  auto nonfinite = [&](auto base, int off, int n) {
    int bad = 0;
    for (int i = tid; i < n; i += 64) { const double v = (double)base[off + i]; if (!(fabs(v) < 1e300)) bad = 1; }
    return (unsigned long long)wave_max(bad);
  };
  auto trace = [&](int cp, unsigned long long bits) {
    if (tid == 0 && st.prof && period < 4) st.prof[(size_t)b * PF_N + 4 * period + cp] = bits | (1ull << 63);
  };
#endif
  // ---- load persistent state (lane-contiguous records) and form the QP data in double.  All global
  //      loads of a block are issued before the first use so their latencies overlap.
  const P<double> gX = mk(st.X + (size_t)b * (N + 1) * NX, (N + 1) * NX, CK_X);
  const P<double> gU = mk(st.U + (size_t)b * N * NU, N * NU, CK_U);
  const int idx = st.idx[b];
  int have = 0, len = 1;
  P<const double> tr = nullptr;
  if (mode & MODE_TRAJ) {
    len = st.tlen[b];
    have = chunk_have(len, idx, N, m.skip);
    tr = mk(st.traj + (size_t)b * m.Tmax * NX, (long)m.Tmax * NX, CK_TRAJ);
  }
  const P<const double> gy = mk((const double*)st.yref + (size_t)b * N * NY, N * NY, CK_YREF);
  const P<const double> gyN = mk((const double*)st.yrefN + (size_t)b * NX, NX, CK_YREFN);
  auto xref = [&](int i, int k) -> double {  // reference of node i (i == N: terminal = chunk row N-1)
    if (mode & MODE_TRAJ) return tr[chunk_row(i < N ? i : N - 1, have, idx, m.skip, len) * NX + k];
    return i < N ? gy[i * NY + k] : gyN[k];
  };
  auto uref = [&](int i, int k) -> double { return (mode & MODE_TRAJ) ? m.uref[k] : gy[i * NY + NX + k]; };
  const P<TQ> gmu = mk(st.mu + (size_t)b * 3 * nb, 3 * nb, CK_MU);
  const P<double> gW = mk(st.w + (size_t)b * NU, NU, CK_W), gRunX = mk(st.run_x ? st.run_x + (size_t)b * NX : nullptr, NX, CK_RUNX);
  const P<const TQ> mKxinv = mk(m.Kxinv, 3L * nb * nb, CK_KXINV), mBasis = mk(m.basis, 3 * nb, CK_BASIS);
  // Every independent global load of the phase is issued before the first use, so the phase costs about one
  // memory round trip (plus the cursor, which the reference rows depend on).
  constexpr int UNR = 5;
  constexpr bool keep_ref = C::N > 0 && (C::N + 1) * NX <= 64 * UNR;   // reference rows stay in registers for the cost
  double xv[UNR], rv[UNR];
  auto load_block = [&](int base) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = base + u * 64 + tid, itc = it < (N + 1) * NX ? it : 0, i = itc / NX;
      xv[u] = gX[itc];
      rv[u] = xref(i, itc - i * NX);
    }
  };
  load_block(0);
  if (mode & MODE_PLANT_FIRST) {
    PF_SER(PF_LOAD);
    if (tid == 0) {
      double x[NX], u[NU];
#pragma unroll
      for (int k = 0; k < NX; ++k) x[k] = gRunX[k];
#pragma unroll
      for (int k = 0; k < NU; ++k) u[k] = gW[k];
      for (int sub = 0; sub < st.run_nsub; ++sub) plant_rk4(m, x, u, st.run_dt);
#pragma unroll
      for (int k = 0; k < NX; ++k) { gRunX[k] = x[k]; D[L.x0 + k] = x[k]; }   // the new plant state IS this period's measurement
    }
    PF_SER(11);
  }
  const bool meas_from_plant = (mode & MODE_PLANT_FIRST) != 0;   // then lane 0 has already put it into LDS (no store -> load hand-over through memory)
  const double xm = (tid < NX && !meas_from_plant) ? mk(st.x_meas + (size_t)b * NX, NX, CK_XMEAS)[tid] : 0.0;
  if (gp) {   // mu -> LDS scratch (shooting records are not live yet); rows of Kx^-1 come straight from L2
    for (int i = tid; i < 3 * nb; i += 64) { S[L.sub + i] = gmu[i]; S[L.basis + i] = mBasis[i]; }
  }
  // U -> LDS, r0 = R (U_i - uref_i), bounds
  for (int it = tid; it < nv; it += 64) {
    const int i = it >> 2, k = it & 3;
    const double u = gU[it];
    D[L.U + it] = u;
    if (!C::GK) {   // (compact layout: these three share the space of the shooting records and are written behind the shooting)
      S[L.r0 + it] = (TQ)(m.h * m.W[NX + k] * (u - uref(i, k)));
      S[L.lb + it] = (TQ)(m.ulb[k] - u);
      S[L.ub + it] = (TQ)(m.uub[k] - u);
    }
  }
  if (tid < NX && !meas_from_plant) D[L.x0 + tid] = xm;
  // the post phase's reads of the persistent state travel with the loads of this phase (three serial global round trips
  // less behind the QP); a free-running launch keeps them in LDS from the second period on (lane 0 updates both copies)
  if ((mode & MODE_POST) && (!C::RUN || period == 0) && tid < NX + 5)
    D[L.pre + tid] = tid < NX ? st.xpp[(size_t)b * NX + tid] : (tid < NX + 4 ? st.stats[(size_t)b * 4 + (tid - NX)] : (double)st.has_prev[b]);
  // X -> LDS and qv = Q_i (X_i - xref_i) in one pass over the record
  for (int base = 0; base < (N + 1) * NX; base += 64 * UNR) {
    if (base > 0) load_block(base);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = base + u * 64 + tid;
      if (it < (N + 1) * NX) {
        const int i = it / NX, k = it - i * NX;
        const double q = i < N ? m.h * m.W[k] : m.We[k];
        D[L.X + it] = xv[u];
        A[L.qv + i * VS + o2i(k)] = st16((TQ)(q * (xv[u] - rv[u])));
      }
    }
  }
  for (int it = tid; it < (N + 1) * 3; it += 64) A[L.qv + (it / 3) * VS + NX + it % 3] = 0;
  if (tid < VS) {   // weights in internal order: stage, terminal, input
    S[L.wq + tid] = tid < NX ? (TQ)(m.h * m.W[i2o(tid)]) : TQ(0);
    S[L.wq + VS + tid] = tid < NX ? (TQ)m.We[i2o(tid)] : TQ(0);
    S[L.wq + 2 * VS + tid] = tid < NU ? (TQ)(m.h * m.W[NX + tid]) : TQ(0);
  }
  if (gp) {
    __syncthreads();
    // alpha = Kx^-1 mu  (the OCP model evaluates k*(v_b) Kx^-1 p, src/gp/RGP.py:250-254)
    for (int i = tid; i < 3 * nb; i += 64) {
      const int d = i / nb, r = i - d * nb;
      const P<const TQ> kr = mKxinv + (d * nb * nb + r * nb);
      const P<TQ> mu = S + (L.sub + d * nb);
      TQ t = 0;
      int k = 0;
      for (; k + 5 <= nb; k += 5) {
        const TQ a0 = kr[k], a1 = kr[k + 1], a2 = kr[k + 2], a3 = kr[k + 3], a4 = kr[k + 4];
        t += a0 * mu[k] + a1 * mu[k + 1] + a2 * mu[k + 2] + a3 * mu[k + 3] + a4 * mu[k + 4];
      }
      for (; k < nb; ++k) t += kr[k] * mu[k];
      S[L.alpha + i] = t;
    }
  }
  __syncthreads();
  PF_STOP(PF_LOAD);
#ifdef MPCQ_TRACE_NAN
  trace(0, nonfinite(D, L.X, (N + 1) * NX) | nonfinite(D, L.U, nv) << 1 | nonfinite(D, L.x0, NX) << 2 | nonfinite(A, L.qv, (N + 1) * VS) << 3 |
               (gp ? nonfinite(S, L.alpha, 3 * nb) : 0ull) << 4 | (unsigned long long)(unsigned)idx << 32);
#endif
  // ---- 1. shooting
  shoot_states<C>(m, D, S, A, L, gp);
  __syncthreads();
  PF_STOP(PF_SHOOT_X);
  shoot_sens<C>(m, S, A, L);
  __syncthreads();
  PF_STOP(PF_SHOOT_S);   // shooting records (union region) are dead from here on
  PF_BUCKET(11);         // (11: between the shooting and the QP)
#ifdef MPCQ_TRACE_NAN
  trace(1, nonfinite(A, L.c, N * VS) | nonfinite(A, L.AB, N * ABS) << 1);
#endif
  if (tid < VS) { S[L.dx + tid] = 0; A[L.AB + N * ABS + tid] = 0; S[L.zb + tid] = 0; }
  for (int it = tid; it < N * VS; it += 64) S[L.vin + it] = 0;
  if (C::GK) {   // r0 = R (U_i - uref_i), bounds: the same expressions as in the load phase, from the iterate in LDS
    for (int it = tid; it < nv; it += 64) {
      const int i = it >> 2, k = it & 3;
      const double u = D[L.U + it];
      S[L.r0 + it] = (TQ)(m.h * m.W[NX + k] * (u - uref(i, k)));
      S[L.lb + it] = (TQ)(m.ulb[k] - u);
      S[L.ub + it] = (TQ)(m.uub[k] - u);
    }
  }
  __syncthreads();
  if (tid < NX) S[L.dx + o2i(tid)] = (TQ)(D[L.x0 + tid] - D[L.X + tid]);   // dx_0 = x_meas - X_0 (lbx = ubx = x_init)
  __syncthreads();
  // ---- 2. QP
  int status = 0;
  const int prev_iter = st.qp_iter[b];
  int work = 0;
  int iters;
  if constexpr (sizeof(TQ) == 4) iters = solve_qp_mixed<C>(m, D, S, A, G, L, &status, prev_iter, &work PF_PASS);
  else iters = solve_qp<C>(m, D, S, A, G, L, &status, prev_iter, &work PF_PASS);
  PF_MARK(15);           // (15: from the return of the QP solve to the full step)
#ifdef MPCQ_TRACE_NAN
  trace(2, nonfinite(S, L.z, nv) | nonfinite(S, L.dx, (N + 1) * VS) << 1 | (unsigned long long)(status & 0xff) << 8 | (unsigned long long)(unsigned)iters << 32);
#endif
  PF_START();
  // ---- 3. full step (iterate accumulated in double).  A step that is not finite (a QP that broke down: only seen
  //      with the fp32 QP on infeasible references) is not taken: the iterate and the control of the previous period
  //      stay, the instance reports MPCQ_SOLVE_NAN and starts the next period from a sound iterate.
  // (float instances hand the QP solution over in double: D[L.dxd], D[L.zd], refined against fp64 residuals)
  auto sol_dx = [&](int k) -> double { if constexpr (sizeof(TQ) == 4) return D[L.dxd + k]; else return (double)S[L.dx + k]; };
  auto sol_z = [&](int k) -> double { if constexpr (sizeof(TQ) == 4) return D[L.zd + k]; else return (double)S[L.z + k]; };
  int unsound = 0;
  for (int it = tid; it < (N + 1) * VS; it += 64) { const double v = sol_dx(it); if (!(fabs(v) < 1e30)) unsound = 1; }
  for (int i = tid; i < nv; i += 64) { const double v = sol_z(i); if (!(fabs(v) < 1e30)) unsound = 1; }
  unsound = wave_max(unsound);
  if (unsound) status = 1;
  else {
    for (int it = tid; it < (N + 1) * NX; it += 64) {
      const int i = it / NX, k = it - i * NX;
      const double v = D[L.X + it] + sol_dx(i * VS + o2i(k));
      D[L.X + it] = v; gX[it] = v;
    }
    for (int i = tid; i < nv; i += 64) {
      double v = D[L.U + i] + sol_z(i);
      v = tmin(tmax(v, m.ulb[i & 3]), m.uub[i & 3]);   // the QP keeps du inside [lb, ub]; removes the rounding of TQ -> double
      D[L.U + i] = v; gU[i] = v;
    }
  }
  __syncthreads();
  // cost at the new iterate (get_cost)
  double cst = 0;
  int bad = 0;
  if (keep_ref) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = u * 64 + tid;
      if (it < (N + 1) * NX) {
        const int i = it / NX, k = it - i * NX;
        const double v = D[L.X + it], e = v - rv[u];
        cst += 0.5 * (i < N ? m.h * m.W[k] : m.We[k]) * e * e;
        if (!(v == v)) bad = 1;
      }
    }
  } else {
    for (int it = tid; it < (N + 1) * NX; it += 64) {
      const int i = it / NX, k = it - i * NX;
      const double v = D[L.X + it], e = v - xref(i, k);
      cst += 0.5 * (i < N ? m.h * m.W[k] : m.We[k]) * e * e;
      if (!(v == v)) bad = 1;
    }
  }
  for (int it = tid; it < nv; it += 64) {
    const double v = D[L.U + it], e = v - uref(it >> 2, it & 3);
    cst += 0.5 * m.h * m.W[NX + (it & 3)] * e * e;
    if (!(v == v)) bad = 1;
  }
  cst = wave_sum(cst);
  bad = wave_max(bad);
  if (bad) status = 1;
  if (tid == 0) { st.cost[b] = cst; st.status[b] = status; st.qp_iter[b] = iters; st.qp_work[b] = work; }
#ifdef MPCQ_TRACE_NAN
  trace(3, (unsigned long long)(status & 0xff) | (unsigned long long)unsound << 8 | (unsigned long long)bad << 9 | (unsigned long long)(unsigned)prev_iter << 32);
#endif
  if (tid < NU) {
    gW[tid] = D[L.U + tid];
    if (st.w_ext) st.w_ext[(size_t)b * NU + tid] = D[L.U + tid];
  }
  PF_STOP(PF_ELEM);
  PF_START();
  if (mode & MODE_POST) {
  // ---- 4. post: nominal prediction, cursor, drag estimate, RGP regress, statistics
  const P<double> vbad = D + (L.x0 + NX);   // [v_body(3), a_drag(3)]
  const P<TQ> gCov = mk(st.C + (size_t)b * 3 * nb * nb, 3L * nb * nb, CK_C);
  const P<double> gXpp = mk(st.xpp + (size_t)b * NX, NX, CK_XPP), gXpred = mk(st.xpred + (size_t)b * NX, NX, CK_XPRED), pre = D + L.pre;
  const bool regress = gp && !(mode & MODE_STATIC_GP);
  if (regress) {   // stage the covariance while lane 0 integrates the nominal model (QP workspace is dead)
    for (int i = tid; i < 3 * nb * nb; i += 64) S[L.rgp + i] = gCov[i];
  }
  PF_SER(PF_POST);
  if (tid == 0) {
    double x[NX], u[NU], xp[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) x[k] = D[L.x0 + k];
#pragma unroll
    for (int k = 0; k < NU; ++k) u[k] = D[L.U + k];
    rk4_nominal(m, x, u, m.dt_pred, xp);
    PF_SER(12);
    // compute_a_drag (src/utils/utils.py:934-950) against the previous step's prediction
    double xq[NX];
    const bool hp = pre[NX + 4] != 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) xq[k] = hp ? pre[k] : x[k];
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
    for (int k = 0; k < NX; ++k) { gXpred[k] = xp[k]; gXpp[k] = xp[k]; pre[k] = xp[k]; }
    st.has_prev[b] = 1; pre[NX + 4] = 1.0;
    st.idx[b] = idx + 1;
    // tracking statistic against the first row of the reference chunk
    double ep = 0, ev = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double a = x[k] - xref(0, k), c = x[7 + k] - xref(0, 7 + k);
      ep += a * a; ev += c * c;
    }
    const P<double> gs = mk(st.stats + (size_t)b * 4, 4, CK_STATS);
    {
      const double s0 = pre[NX] + ep, s1 = pre[NX + 1] + ev, s2 = pre[NX + 2] + 1, s3 = tmax(pre[NX + 3], ep);
      gs[0] = s0; gs[1] = s1; gs[2] = s2; gs[3] = s3;
      pre[NX] = s0; pre[NX + 1] = s1; pre[NX + 2] = s2; pre[NX + 3] = s3;
    }
    // trajectory finished (src/mpc_controller_node.py:374, evaluated after idx_traj += 1): the cursor stands on the last
    // row and the quadrotor is within EPSILON_TRAJECTORY_FINISHED of the first row of this step's chunk
    if ((mode & MODE_TRAJ) && idx + 2 == len && sqrt(ep) < m.finish_r) st.finished[b] = 1;
  }
  __syncthreads();
  PF_SER(13);
  if (regress) rgp_regress<C>(m, S, L, gmu, gCov, vbad, vbad + 3, true);
  PF_SER(14);
  }
  if (C::RUN) {   // the plant produces the next measurement (plant_kernel of the lockstep path)
    if (tid == 0) {
      double x[NX], u[NU];
#pragma unroll
      for (int k = 0; k < NX; ++k) x[k] = gRunX[k];
#pragma unroll
      for (int k = 0; k < NU; ++k) u[k] = D[L.U + k];
      for (int sub = 0; sub < st.run_nsub; ++sub) plant_rk4(m, x, u, st.run_dt);
#pragma unroll
      for (int k = 0; k < NX; ++k) gRunX[k] = x[k];
    }
  }
  __syncthreads();
  }   // periods
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
  const Lds L = lds_layout(m.N, m.nb, m.gab, sizeof(TQ) == 4);
#ifdef MPCQ_CHECKED
  if (threadIdx.x == 0) *reinterpret_cast<int**>(smem_raw) = st.chk;
  __syncthreads();
#endif
  const P<TQ> S = mk(reinterpret_cast<TQ*>(smem_raw + CK_HDR + L.dbytes), L.qtotal, CK_LDS_Q);
  const int b = blockIdx.x, nb = m.nb;
  rgp_regress<Cfg<TQ, false>>(m, S, L, mk(st.mu + (size_t)b * 3 * nb, 3 * nb, CK_MU), mk(st.C + (size_t)b * 3 * nb * nb, 3L * nb * nb, CK_C),
                              mk(vb + (size_t)b * 3, 3, CK_XMEAS), mk(ad + (size_t)b * 3, 3, CK_XMEAS));
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

// publish_control_gazebo (src/mpc_controller_node.py:600-612): rotor_thrusts = w T_max / m, collective_thrust =
// sum(w) T_max / m, body rates = x_opt[1, 10:13]; same operation order as the numpy expressions
template <typename TQ>
__global__ void command_kernel(const DevModel<TQ> m, const double* w, const double* X, double* rotor, double* coll, double* rates, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double wk = w[(size_t)b * NU + k];
    rotor[(size_t)b * NU + k] = wk * m.tmax / m.mass;
    s += wk;
  }
  coll[b] = s * m.tmax / m.mass;
#pragma unroll
  for (int k = 0; k < 3; ++k) rates[(size_t)b * 3 + k] = X[(size_t)b * (m.N + 1) * NX + NX + 10 + k];
}

// get_reference_chunk at the current cursor, with the same row selection the step kernel uses: out [B][N][13]
template <typename TQ>
__global__ void chunk_kernel(const DevModel<TQ> m, const double* traj, const int* tlen, const int* idx, double* out) {
  const int b = blockIdx.x;
  const int len = tlen[b], id = idx[b], have = chunk_have(len, id, m.N, m.skip);
  for (int it = threadIdx.x; it < m.N * NX; it += blockDim.x) {
    const int j = it / NX, k = it - j * NX;
    out[(size_t)b * m.N * NX + it] = traj[((size_t)b * m.Tmax + chunk_row(j, have, id, m.skip, len)) * NX + k];
  }
}

// reduce per-instance statistics to 5 numbers (sum, sum, sum, max, #failed)
static __global__ void stats_kernel(const double* stats, const int* status, int B, double* out5) {
  double (*sh)[256] = reinterpret_cast<double (*)[256]>(smem_raw);  // [5][256]
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    a0 += stats[(size_t)b * 4]; a1 += stats[(size_t)b * 4 + 1]; a2 += stats[(size_t)b * 4 + 2];
    const double mx = stats[(size_t)b * 4 + 3];
    a3 = a3 > mx ? a3 : mx;
    a4 += (status[b] & 7) != 0 ? 1.0 : 0.0;   // MPCQ_SOLVE_LOW_ACCURACY (8) is a warning, not a failed solve
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


// ------------------------------------------------------------------ launch order of a lockstep period (large batches)
// A batch larger than the device holds at once runs as a stream of workgroups, and the launch ends with its last one: the
// expensive quadrotors have to start first (longest-processing-time-first list scheduling).  What this period's solve will
// cost is predicted from the previous one (qp_iter: passes + interior-point iterations; a quadrotor that fell back or carries
// the flip mark will do so again more often than not).  Workgroup p runs quadrotor order[p]; the permutation stays inside the
// classes p mod 8 -- workgroups are dealt round-robin over the 8 XCDs, so every XCD gets its own share of the expensive ones
// first and a quadrotor keeps its XCD -- each class in descending predicted cost (stable counting sort, 16 bins).
// One workgroup of 256 lanes per class; dynamic LDS: 16 x 256 counters + 16 bin bases.
constexpr int ORD_BINS = 16, ORD_THREADS = 256, ORD_CLASSES = 8;
constexpr size_t ORD_LDS = (size_t)(ORD_BINS * ORD_THREADS + ORD_BINS) * sizeof(int);
__host__ __device__ inline int order_bin(int q) {   // bin 0 = most expensive
  if (q < 0) q = 0;   // (not a value the solver writes; whatever a restored checkpoint holds, the result is a bin)
  const int total = q % 1000, marked = (q / 1000) % 100 != 0;   // fallback solve (x 1000) or flip mark (x 10000)
  int cost = q == 0 ? ORD_BINS - 1 : (total < ORD_BINS - 1 ? total : ORD_BINS - 1);   // cold start: interior point from scratch
  if (marked && cost < 8) cost = 8;
  return ORD_BINS - 1 - cost;
}
// (qp_iter, order: the segment of the group this launch sorts; first: its first quadrotor -- the entries written are global indices)
static __global__ void __launch_bounds__(ORD_THREADS) order_kernel(const int* qp_iter, int B, int* order, int first) {
  int* cnt = reinterpret_cast<int*>(smem_raw);          // [ORD_BINS][ORD_THREADS]
  int* base = cnt + ORD_BINS * ORD_THREADS;             // [ORD_BINS]
  const int x = blockIdx.x, t = threadIdx.x;
  const int n = (B - x + ORD_CLASSES - 1) / ORD_CLASSES;      // members of the class: b = 8 j + x < B
  const int chunk = (n + ORD_THREADS - 1) / ORD_THREADS;
  const int j0 = t * chunk < n ? t * chunk : n, j1 = j0 + chunk < n ? j0 + chunk : n;
  for (int k = 0; k < ORD_BINS; ++k) cnt[k * ORD_THREADS + t] = 0;
  for (int j = j0; j < j1; ++j) cnt[order_bin(qp_iter[ORD_CLASSES * j + x]) * ORD_THREADS + t] += 1;   // own column: no race
  __syncthreads();
  if (t < ORD_BINS) {   // exclusive scan of bin t over the lanes
    int run = 0;
    for (int u = 0; u < ORD_THREADS; ++u) { const int c = cnt[t * ORD_THREADS + u]; cnt[t * ORD_THREADS + u] = run; run += c; }
    base[t] = run;
  }
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < ORD_BINS; ++k) { const int c = base[k]; base[k] = run; run += c; }
  }
  __syncthreads();
  for (int j = j0; j < j1; ++j) {
    const int k = order_bin(qp_iter[ORD_CLASSES * j + x]);
    const int pos = base[k] + cnt[k * ORD_THREADS + t];
    cnt[k * ORD_THREADS + t] += 1;
    order[ORD_CLASSES * pos + x] = first + ORD_CLASSES * j + x;
  }
}

}  // namespace mpcq
