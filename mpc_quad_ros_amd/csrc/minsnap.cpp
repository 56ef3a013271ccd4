// minsnap.cpp — host-side minimum-snap trajectory generator through waypoints (libmpcq_traj.so, plain C++, no HIP).
//
// Replaces what the reference obtains by shelling out to its prebuilt `genTrajectory` binary
// (src/trajectory_generation/TrajectoryGenerator.py:177-191: waypoints.csv -> polynomial_representation.csv under
// --v_max / --a_max): per axis a chain of 7th-order polynomials (8 coefficients per segment, the format of
// src/trajectory_generation/uav_trajectory.py:116-129) through the waypoints that minimises the integral of the squared
// snap, with position fixed at every waypoint, velocity / acceleration / jerk continuous at the interior ones (and free
// there) and zero at both ends; segment times from the distance / v_max / a_max ramp estimate, then stretched or shrunk
// uniformly until the sampled speed and acceleration sit on their limits.  The binary itself (mav_trajectory_generation +
// nlopt, no source in the reference tree) cannot run here.  What its outputs say (tests/test_minsnap.py, round 6, on the three
// logged references through waypoints/user_defined_waypoints.csv): they ARE chains of 7th-order pieces (a free per-piece fit leaves
// the 6-decimal rounding of the samples), through the waypoints and C^3 to what the "%.6f" coefficients of its CSV allow -- i.e.
// points of the family (segment times T, free vertex derivatives d_P) of mpcq_minsnap_from_derivatives below, which reproduces them
// to that rounding; their d_P are NOT the optimum of the linear stage at their own T (off by up to 40 % of the velocity scale): the
// binary's nonlinear stage moved times AND derivatives (an early-stopped Subplex run over both).  So this generator is the same
// trajectory FAMILY, not the same trajectories.
//
// The unconstrained formulation: with d = (p, v, a, j) at both ends of a segment, coefficients c = A(T)^-1 d and cost
// c' Q(T) c, so the total cost is a quadratic form in the vertex derivatives; the free ones (v, a, j at interior waypoints)
// follow from one small linear solve per axis.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr int NC = 8;   // coefficients per segment and axis
constexpr int ND = 4;   // derivatives held at a vertex: p, v, a, j

// dense Gaussian elimination with partial pivoting, n right-hand sides in B (row-major n x m); returns false if singular
bool solve_dense(std::vector<double>& A, std::vector<double>& B, int n, int m) {
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(A[r * n + c]) > std::fabs(A[p * n + c])) p = r;
    if (A[p * n + c] == 0.0) return false;
    if (p != c) {
      for (int k = 0; k < n; ++k) std::swap(A[p * n + k], A[c * n + k]);
      for (int k = 0; k < m; ++k) std::swap(B[p * m + k], B[c * m + k]);
    }
    const double inv = 1.0 / A[c * n + c];
    for (int r = c + 1; r < n; ++r) {
      const double f = A[r * n + c] * inv;
      if (f == 0.0) continue;
      for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
      for (int k = 0; k < m; ++k) B[r * m + k] -= f * B[c * m + k];
    }
  }
  for (int c = n - 1; c >= 0; --c) {
    for (int k = 0; k < m; ++k) {
      double s = B[c * m + k];
      for (int j = c + 1; j < n; ++j) s -= A[c * n + j] * B[j * m + k];
      B[c * m + k] = s / A[c * n + c];
    }
  }
  return true;
}

// rows: derivative r (0..3) of sum_i c_i t^i at t, as a row over the coefficients
void deriv_row(double t, int r, double* row) {
  for (int i = 0; i < NC; ++i) {
    if (i < r) { row[i] = 0; continue; }
    double f = 1;
    for (int k = 0; k < r; ++k) f *= (i - k);
    row[i] = f * std::pow(t, i - r);
  }
}

// M = A(T)^-T Q(T) A(T)^-1: snap cost of one segment as a quadratic form in d = [p v a j](0), [p v a j](T); also A^-1
void segment_cost(double T, double M[2 * ND][2 * ND], double Ainv[NC][NC], int order = 4) {
  std::vector<double> A(NC * NC), I(NC * NC, 0.0);
  for (int r = 0; r < ND; ++r) {
    deriv_row(0.0, r, &A[r * NC]);
    deriv_row(T, r, &A[(ND + r) * NC]);
  }
  for (int i = 0; i < NC; ++i) I[i * NC + i] = 1.0;
  solve_dense(A, I, NC, NC);   // I <- A^-1
  for (int i = 0; i < NC; ++i)
    for (int j = 0; j < NC; ++j) Ainv[i][j] = I[i * NC + j];
  double Q[NC][NC] = {};
  auto fall = [&](int i) { double f = 1; for (int k = 0; k < order; ++k) f *= (i - k); return f; };   // i! / (i - order)!
  for (int i = order; i < NC; ++i)
    for (int j = order; j < NC; ++j) Q[i][j] = fall(i) * fall(j) * std::pow(T, i + j - 2 * order + 1) / (i + j - 2 * order + 1);
  for (int a = 0; a < 2 * ND; ++a)
    for (int b = 0; b < 2 * ND; ++b) {
      double s = 0;
      for (int i = order; i < NC; ++i)
        for (int j = order; j < NC; ++j) s += Ainv[i][a] * Q[i][j] * Ainv[j][b];
      M[a][b] = s;
    }
}

// one axis: positions p[0..n-1] at the vertices, segment times T[0..n-2] -> coefficients coef[seg][8]
bool solve_axis(const double* p, int n, const double* T, double* coef, int order = 4) {
  const int ns = n - 1, nfree = 3 * (n - 2);
  std::vector<double> R((size_t)nfree * nfree, 0.0), rhs(nfree, 0.0);
  std::vector<std::vector<double>> Ainvs(ns, std::vector<double>(NC * NC));
  // vertex derivative index: fixed (value known) or free (index into the unknowns)
  auto free_idx = [&](int vertex, int r) { return (vertex == 0 || vertex == n - 1 || r == 0) ? -1 : 3 * (vertex - 1) + (r - 1); };
  auto fixed_val = [&](int vertex, int r) { return r == 0 ? p[vertex] : 0.0; };
  for (int s = 0; s < ns; ++s) {
    double M[2 * ND][2 * ND], Ai[NC][NC];
    segment_cost(T[s], M, Ai, order);
    for (int i = 0; i < NC; ++i)
      for (int j = 0; j < NC; ++j) Ainvs[s][i * NC + j] = Ai[i][j];
    for (int a = 0; a < 2 * ND; ++a) {
      const int va = s + a / ND, ra = a % ND, fa = free_idx(va, ra);
      if (fa < 0) continue;
      for (int b = 0; b < 2 * ND; ++b) {
        const int vb = s + b / ND, rb = b % ND, fb = free_idx(vb, rb);
        if (fb >= 0) R[(size_t)fa * nfree + fb] += M[a][b];
        else rhs[fa] -= M[a][b] * fixed_val(vb, rb);
      }
    }
  }
  if (nfree > 0 && !solve_dense(R, rhs, nfree, 1)) return false;
  for (int s = 0; s < ns; ++s) {
    double d[2 * ND];
    for (int a = 0; a < 2 * ND; ++a) {
      const int v = s + a / ND, r = a % ND, f = free_idx(v, r);
      d[a] = f >= 0 ? rhs[f] : fixed_val(v, r);
    }
    for (int i = 0; i < NC; ++i) {
      double c = 0;
      for (int a = 0; a < 2 * ND; ++a) c += Ainvs[s][i * NC + a] * d[a];
      coef[s * NC + i] = c;
    }
  }
  return true;
}

// one axis with GIVEN free derivatives dfree[3 (n - 2)] (v, a, j at the interior vertices): coefficients and cost d' M d
void build_axis(const double* p, int n, const double* T, const double* dfree, double* coef, int order, double* cost) {
  const int ns = n - 1;
  double J = 0;
  for (int s = 0; s < ns; ++s) {
    double M[2 * ND][2 * ND], Ai[NC][NC], d[2 * ND];
    segment_cost(T[s], M, Ai, order);
    for (int a = 0; a < 2 * ND; ++a) {
      const int v = s + a / ND, r = a % ND;
      d[a] = r == 0 ? p[v] : ((v == 0 || v == n - 1) ? 0.0 : dfree[3 * (v - 1) + (r - 1)]);
    }
    for (int i = 0; i < NC; ++i) {
      double c = 0;
      for (int a = 0; a < 2 * ND; ++a) c += Ai[i][a] * d[a];
      coef[s * NC + i] = c;
    }
    for (int a = 0; a < 2 * ND; ++a)
      for (int b = 0; b < 2 * ND; ++b) J += d[a] * M[a][b] * d[b];
  }
  *cost = J;
}

// largest speed and acceleration magnitude over the trajectory, sampled every dt
void limits(const double* coef, const double* T, int ns, double dt, double* vmax, double* amax) {
  double vm = 0, am = 0;
  for (int s = 0; s < ns; ++s) {
    const int steps = std::max(2, (int)std::ceil(T[s] / dt));
    for (int k = 0; k <= steps; ++k) {
      const double t = T[s] * k / steps;
      double v[3], a[3];
      for (int ax = 0; ax < 3; ++ax) {
        const double* c = coef + ((size_t)s * 4 + ax) * NC;
        double vv = 0, aa = 0;
        for (int i = NC - 1; i >= 1; --i) vv = vv * t + i * c[i];
        for (int i = NC - 1; i >= 2; --i) aa = aa * t + (double)i * (i - 1) * c[i];
        v[ax] = vv; a[ax] = aa;
      }
      vm = std::max(vm, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
      am = std::max(am, std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]));
    }
  }
  *vmax = vm; *amax = am;
}

}  // namespace

extern "C" {

int mpcq_minsnap_solve_order(const double* wp, int32_t n, const double* T, int32_t derivative_to_optimize, double* pieces);
int mpcq_minsnap_generate_order(const double* wp, int32_t n, double v_max, double a_max, int32_t derivative_to_optimize, double* pieces);
int mpcq_minsnap_from_derivatives(const double* wp, int32_t n, const double* T, const double* d_free, int32_t derivative_to_optimize, double* pieces, double* cost);

// Segment-time estimate from distance and the limits (velocity ramp: t = 2 d / v_max (1 + 6.5 v_max / a_max exp(-2 d / v_max))).
int mpcq_minsnap_estimate_times(const double* wp, int32_t n, double v_max, double a_max, double* T) {
  if (!wp || !T || n < 2 || !(v_max > 0) || !(a_max > 0)) return -1;
  for (int s = 0; s < n - 1; ++s) {
    double d2 = 0;
    for (int k = 0; k < 3; ++k) d2 += (wp[(s + 1) * 3 + k] - wp[s * 3 + k]) * (wp[(s + 1) * 3 + k] - wp[s * 3 + k]);
    const double d = std::sqrt(d2);
    T[s] = std::max(1e-3, 2.0 * d / v_max * (1.0 + 6.5 * v_max / a_max * std::exp(-2.0 * d / v_max)));
  }
  return 0;
}

// Minimum-snap polynomials for given segment times.  wp [n,3], T [n-1] -> pieces [n-1, 33] in the reference's CSV row
// layout (duration, x^0..x^7, y^0..y^7, z^0..z^7, yaw^0..yaw^7; yaw = 0).
int mpcq_minsnap_solve(const double* wp, int32_t n, const double* T, double* pieces) { return mpcq_minsnap_solve_order(wp, n, T, 4, pieces); }

// The same linear solve for the derivative the cost penalises: 4 = snap, 3 = jerk (what the reference's binary is built with:
// mav_trajectory_generation::PolynomialOptimizationNonLinear<8>, derivative_to_optimize = JERK -- DESIGN.md section 6.1), 2 = acceleration.
// This is PolynomialOptimization<8>::solveLinear of that library (Richter, Bry, Roy 2013; Burri et al. 2015): vertices made with
// makeStartOrEnd(position, derivative_to_optimize) at both ends (derivatives 1..3 zero), position-only in between, the free vertex
// derivatives from d_P = -R_PP^-1 R_FP^T d_F -- solve_axis above.
int mpcq_minsnap_solve_order(const double* wp, int32_t n, const double* T, int32_t derivative_to_optimize, double* pieces) {
  if (!wp || !T || !pieces || n < 2 || derivative_to_optimize < 2 || derivative_to_optimize > 4) return -1;
  const int ns = n - 1;
  for (int s = 0; s < ns; ++s)
    if (!(T[s] > 0)) return -1;
  std::vector<double> p(n), c((size_t)ns * NC);
  for (int s = 0; s < ns; ++s) {
    pieces[(size_t)s * 33] = T[s];
    for (int i = 0; i < NC; ++i) pieces[(size_t)s * 33 + 25 + i] = 0.0;
  }
  for (int ax = 0; ax < 3; ++ax) {
    for (int v = 0; v < n; ++v) p[v] = wp[v * 3 + ax];
    if (!solve_axis(p.data(), n, T, c.data(), derivative_to_optimize)) return -2;
    for (int s = 0; s < ns; ++s)
      for (int i = 0; i < NC; ++i) pieces[(size_t)s * 33 + 1 + ax * NC + i] = c[(size_t)s * NC + i];
  }
  return 0;
}

// The map the reference generator's NONLINEAR stage evaluates at every iterate (PolynomialOptimizationNonLinear<8>: nlopt varies the
// segment times and the free vertex derivatives d_P; the polynomials and the cost follow from them): wp [n,3], T [n-1], d_free
// [n-2][3 axes][3: velocity, acceleration, jerk] at the interior waypoints (both ends at rest) -> pieces [n-1,33] and, if cost is not
// NULL, the integral of the squared derivative_to_optimize summed over the axes (solve_order returns the d_free that minimise it for
// the given T).  The reference's logged trajectories are points of this family (tests/test_minsnap.py).
int mpcq_minsnap_from_derivatives(const double* wp, int32_t n, const double* T, const double* d_free, int32_t derivative_to_optimize, double* pieces, double* cost) {
  if (!wp || !T || !pieces || n < 2 || (n > 2 && !d_free) || derivative_to_optimize < 2 || derivative_to_optimize > 4) return -1;
  const int ns = n - 1;
  for (int s = 0; s < ns; ++s)
    if (!(T[s] > 0)) return -1;
  std::vector<double> p(n), c((size_t)ns * NC), df((size_t)3 * (n > 2 ? n - 2 : 0));
  double total = 0;
  for (int s = 0; s < ns; ++s) {
    pieces[(size_t)s * 33] = T[s];
    for (int i = 0; i < NC; ++i) pieces[(size_t)s * 33 + 25 + i] = 0.0;
  }
  for (int ax = 0; ax < 3; ++ax) {
    for (int v = 0; v < n; ++v) p[v] = wp[v * 3 + ax];
    for (int v = 0; v < n - 2; ++v)
      for (int r = 0; r < 3; ++r) df[3 * v + r] = d_free[((size_t)v * 3 + ax) * 3 + r];
    double J = 0;
    build_axis(p.data(), n, T, df.data(), c.data(), derivative_to_optimize, &J);
    total += J;
    for (int s = 0; s < ns; ++s)
      for (int i = 0; i < NC; ++i) pieces[(size_t)s * 33 + 1 + ax * NC + i] = c[(size_t)s * NC + i];
  }
  if (cost) *cost = total;
  return 0;
}

// The LINEAR stage of the reference's generator, as published: segment times from estimateSegmentTimes(vertices, v_max, a_max) =
// estimateSegmentTimesNfabian with its constant 6.5 (mpcq_minsnap_estimate_times), then the linear solve for derivative_to_optimize.
// The binary continues from exactly this point with nlopt's Subplex over times and free derivatives, stopped early at loose
// tolerances -- that part is not reproducible; this part is, and tests/test_minsnap.py documents how far it
// is from the logged references per waypoint file.  No scaling onto the limits: peak speed / acceleration are what they are.
int mpcq_minsnap_linear(const double* wp, int32_t n, double v_max, double a_max, int32_t derivative_to_optimize, double* pieces) {
  if (n < 2) return -1;
  std::vector<double> T(n - 1);
  if (mpcq_minsnap_estimate_times(wp, n, v_max, a_max, T.data())) return -1;
  return mpcq_minsnap_solve_order(wp, n, T.data(), derivative_to_optimize, pieces);
}

// Full generator: estimate the times, solve, then scale all times by one factor (bisection) so that the sampled peak
// speed / acceleration are within v_max / a_max with the tighter of the two limits reached.  pieces [n-1, 33].
int mpcq_minsnap_generate(const double* wp, int32_t n, double v_max, double a_max, double* pieces) { return mpcq_minsnap_generate_order(wp, n, v_max, a_max, 4, pieces); }
// ... with the cost on derivative_to_optimize (3: the jerk cost of the reference's binary, on the time proportions of its own estimate)
int mpcq_minsnap_generate_order(const double* wp, int32_t n, double v_max, double a_max, int32_t derivative_to_optimize, double* pieces) {
  if (n < 2 || derivative_to_optimize < 2 || derivative_to_optimize > 4) return -1;
  const int ns = n - 1;
  std::vector<double> T0(ns), T(ns), coef((size_t)ns * 4 * NC);
  if (mpcq_minsnap_estimate_times(wp, n, v_max, a_max, T0.data())) return -1;
  auto violation = [&](double scale) {   // > 1: over a limit
    for (int s = 0; s < ns; ++s) T[s] = T0[s] * scale;
    if (mpcq_minsnap_solve_order(wp, n, T.data(), derivative_to_optimize, pieces)) return 1e30;
    for (int s = 0; s < ns; ++s)
      for (int ax = 0; ax < 4; ++ax) std::memcpy(&coef[((size_t)s * 4 + ax) * NC], &pieces[(size_t)s * 33 + 1 + ax * NC], NC * sizeof(double));
    double vm, am;
    limits(coef.data(), T.data(), ns, 0.01, &vm, &am);
    return std::max(vm / v_max, am / a_max);
  };
  // time scaling by s divides speeds by s and accelerations by s^2 for FIXED derivatives; with re-solved free derivatives
  // the relation is only approximately that, hence the bisection on the true sampled peaks
  double lo = 0.05, hi = 1.0;
  while (violation(hi) > 1.0) { hi *= 1.6; if (hi > 1e3) return -3; }
  if (violation(lo) <= 1.0) hi = lo;
  else
    for (int it = 0; it < 40 && hi - lo > 1e-4 * hi; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (violation(mid) > 1.0) lo = mid; else hi = mid;
    }
  return violation(hi) <= 1.0 ? 0 : -3;
}

// Sampling of the pieces into the 13-state reference, the chain save_evals_csv -> load_trajectory of the reference
// (src/trajectory_generation/TrajectoryGenerator.py:208-244): t_k = k dt for t_k < total duration, piece lookup by the running
// sum of the durations (uav_trajectory.py:146-150), Horner evaluation with the highest power first (uav_trajectory.py:22-28),
// positions and velocities rounded to the CSV's 6 decimals, q = [1,0,0,0], body rates 0.  Same operations in the same order as
// mpc_quad_ros_amd.trajectories.sample_polynomial_trajectory (bit-identical; built with -ffp-contract=off).
// x [cap, 13]; returns the number of rows, or -1 if cap is too small / bad arguments.
int mpcq_minsnap_sample(const double* pieces, int32_t nseg, double dt, double* x, int32_t cap) {
  if (!pieces || !x || nseg < 1 || !(dt > 0)) return -1;
  double total = 0;
  std::vector<double> ends(nseg);
  for (int s = 0; s < nseg; ++s) { total = total + pieces[(size_t)s * 33]; ends[s] = total; }
  const int T = (int)std::ceil(total / dt);      // len(np.arange(0, total, dt))
  if (T > cap) return -1;
  int seg = 0;
  for (int k = 0; k < T; ++k) {
    const double t = k * dt;
    while (seg < nseg - 1 && !(t < ends[seg])) ++seg;      // first piece whose end lies beyond t
    const double tl = t - (seg > 0 ? ends[seg - 1] : 0.0);
    double* row = x + (size_t)k * 13;
    for (int i = 0; i < 13; ++i) row[i] = 0.0;
    row[3] = 1.0;
    for (int a = 0; a < 3; ++a) {
      const double* c = pieces + (size_t)seg * 33 + 1 + 8 * a;
      double p = 0.0, v = 0.0;
      for (int i = 0; i < 8; ++i) p = p * tl + c[7 - i];
      for (int i = 0; i < 7; ++i) v = v * tl + (7 - i) * c[7 - i];
      row[a] = std::nearbyint(p * 1e6) / 1e6;            // np.round(., 6): rint(x * 1e6) / 1e6, ties to even
      row[7 + a] = std::nearbyint(v * 1e6) / 1e6;
    }
  }
  return T;
}

// The reference's polynomial CSV (uav_trajectory.Trajectory.savecsv, src/trajectory_generation/uav_trajectory.py:116-129):
// header line, then per segment 33 numbers with "%.6f".
int mpcq_minsnap_write_csv(const char* path, const double* pieces, int32_t nseg) {
  FILE* f = std::fopen(path, "w");
  if (!f) return -1;
  std::fprintf(f, "# duration,x^0,x^1,x^2,x^3,x^4,x^5,x^6,x^7,y^0,y^1,y^2,y^3,y^4,y^5,y^6,y^7,z^0,z^1,z^2,z^3,z^4,z^5,z^6,z^7,"
                  "yaw^0,yaw^1,yaw^2,yaw^3,yaw^4,yaw^5,yaw^6,yaw^7\n");
  for (int s = 0; s < nseg; ++s)
    for (int i = 0; i < 33; ++i) std::fprintf(f, "%.6f%s", pieces[(size_t)s * 33 + i], i == 32 ? "\n" : ",");
  std::fclose(f);
  return 0;
}

}  // extern "C"
