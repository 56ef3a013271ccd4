/* mpcq_traj.h — C ABI of libmpcq_traj.so: minimum-snap reference trajectories through waypoints (host code, C++).
 *
 * Upstream of the control-step path: the reference writes random waypoints to a CSV and shells out to its
 * prebuilt `genTrajectory` binary (src/trajectory_generation/TrajectoryGenerator.py:133-191), which returns 7th-order
 * polynomial segments in the CSV layout of uav_trajectory.Trajectory (src/trajectory_generation/uav_trajectory.py:116-129);
 * these entry points produce that layout.  Sampling to the 13-state reference stays where the reference has it, in Python
 * (mpc_quad_ros_amd/trajectories.py: sample_polynomial_trajectory = save_evals_csv + load_trajectory,
 * TrajectoryGenerator.py:208-244).
 *
 * All arrays are caller-owned, C-contiguous float64.  wp [n,3] waypoints (the first is the start point), T [n-1] segment
 * durations, pieces [n-1,33] rows (duration, x^0..x^7, y^0..y^7, z^0..z^7, yaw^0..yaw^7).  Return 0 on success,
 * negative on bad arguments (-1), a singular system (-2) or limits that cannot be met (-3).
 */
#ifndef MPCQ_TRAJ_H
#define MPCQ_TRAJ_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ramp estimate of the segment times from distances and the limits --v_max / --a_max of the reference's command line */
int mpcq_minsnap_estimate_times(const double* wp, int32_t n, double v_max, double a_max, double* T);
/* minimum-snap polynomials for given times: positions fixed at the waypoints, v / a / jerk continuous at interior
 * waypoints, zero at both ends */
int mpcq_minsnap_solve(const double* wp, int32_t n, const double* T, double* pieces);
/* the same linear solve with the cost on another derivative: 4 snap (= mpcq_minsnap_solve), 3 jerk (what the reference's genTrajectory is
 * built with), 2 acceleration -- PolynomialOptimization<8>::solveLinear of mav_trajectory_generation */
int mpcq_minsnap_solve_order(const double* wp, int32_t n, const double* T, int32_t derivative_to_optimize, double* pieces);
/* the linear stage of the reference's generator as published: estimateSegmentTimes (Nfabian, constant 6.5) + solveLinear, no scaling onto
 * the limits (the binary's nonlinear stage, an early-stopped Subplex run, is not reproducible: DESIGN.md section 6.1) */
int mpcq_minsnap_linear(const double* wp, int32_t n, double v_max, double a_max, int32_t derivative_to_optimize, double* pieces);
/* Since round 6.  Pieces (and cost) for GIVEN segment times and free vertex derivatives: the map the reference generator's nonlinear stage
 * (mav_trajectory_generation::PolynomialOptimizationNonLinear<8>, behind src/trajectory_generation/TrajectoryGenerator.py:177-191) evaluates
 * at every iterate of its optimiser.  d_free [n-2][3 axes][3] = velocity, acceleration, jerk at the interior waypoints (both ends at rest);
 * cost (may be NULL) = the integral of the squared derivative_to_optimize over the trajectory, summed over x, y, z.  mpcq_minsnap_solve_order
 * returns the pieces of the d_free that minimise that cost for the given T; the trajectories the reference logged are other points of the
 * same family (tests/test_minsnap.py). */
int mpcq_minsnap_from_derivatives(const double* wp, int32_t n, const double* T, const double* d_free, int32_t derivative_to_optimize,
                                  double* pieces, double* cost);
/* estimate + solve + uniform time scaling until the sampled peak speed / acceleration meet v_max / a_max */
int mpcq_minsnap_generate(const double* wp, int32_t n, double v_max, double a_max, double* pieces);
/* ... with the cost on derivative_to_optimize (mpcq_minsnap_generate = 4) */
int mpcq_minsnap_generate_order(const double* wp, int32_t n, double v_max, double a_max, int32_t derivative_to_optimize, double* pieces);
/* pieces -> sampled 13-state reference x [cap,13] every dt (save_evals_csv + load_trajectory, TrajectoryGenerator.py:208-244:
 * 6-decimal rounding, q = [1,0,0,0], rates 0); returns the number of rows or -1 (cap too small / bad arguments) */
int mpcq_minsnap_sample(const double* pieces, int32_t nseg, double dt, double* x, int32_t cap);
/* polynomial_representation.csv in the reference's format ("%.6f", header line) */
int mpcq_minsnap_write_csv(const char* path, const double* pieces, int32_t nseg);

#ifdef __cplusplus
}
#endif
#endif /* MPCQ_TRAJ_H */
