"""Minimum-snap reference generator (csrc/minsnap.cpp behind include/mpcq_traj.h).  The reference's
own generator is a prebuilt binary that cannot run here, so these tests pin the generator's defining properties:
interpolation of the waypoints, rest at both ends, continuity up to jerk, optimality of the snap cost against feasible
perturbations, the v/a limits with the tighter one reached, the reference's CSV format (read back by a restatement of
uav_trajectory.Trajectory.loadcsv) and the sampler chain shared with the reference-generated poly vectors."""
import ctypes
import os
import re

import numpy as np
import pytest

from mpc_quad_ros_amd import trajectories as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def polyval(c, t, der=0):
    c = np.array(c, dtype=float)
    for _ in range(der):
        c = np.array([i * c[i] for i in range(1, len(c))])
    return sum(ci * t ** i for i, ci in enumerate(c))


def snap_cost(pieces):
    J = 0.0
    for row in pieces:
        T = row[0]
        for a in range(3):
            c = row[1 + 8 * a:9 + 8 * a]
            s = np.array([i * (i - 1) * (i - 2) * (i - 3) * c[i] for i in range(4, 8)])     # snap = sum s_k t^k, k = i - 4
            for i in range(4):
                for j in range(4):
                    J += s[i] * s[j] * T ** (i + j + 1) / (i + j + 1)
    return J


def test_header_and_library_agree():
    hdr = open(os.path.join(ROOT, "include", "mpcq_traj.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(mpcq_minsnap_[a-z0-9_]+)\s*\(", hdr)))
    assert names == ["mpcq_minsnap_estimate_times", "mpcq_minsnap_from_derivatives", "mpcq_minsnap_generate", "mpcq_minsnap_generate_order", "mpcq_minsnap_linear", "mpcq_minsnap_sample",
                     "mpcq_minsnap_solve", "mpcq_minsnap_solve_order", "mpcq_minsnap_write_csv"]
    lib = ctypes.CDLL(os.path.join(ROOT, "mpc_quad_ros_amd", "libmpcq_traj.so"))
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_interpolation_continuity_and_rest(seed):
    rng = np.random.default_rng(seed)
    wp = np.vstack([[0, 0, 3.0], rng.uniform(-5, 5, (4, 3)) + [0, 0, 7.5]])
    T = rng.uniform(0.8, 2.5, 4)
    P = tr.minsnap_solve(wp, T)
    assert np.array_equal(P[:, 0], T) and not P[:, 25:].any()          # yaw polynomial identically zero
    for s in range(4):
        for a in range(3):
            c = P[s, 1 + 8 * a:9 + 8 * a]
            assert abs(polyval(c, 0.0) - wp[s, a]) < 1e-9 and abs(polyval(c, T[s]) - wp[s + 1, a]) < 1e-8
            if s < 3:      # velocity, acceleration, jerk continuous at interior waypoints
                cn = P[s + 1, 1 + 8 * a:9 + 8 * a]
                for d in (1, 2, 3):
                    assert abs(polyval(c, T[s], d) - polyval(cn, 0.0, d)) < 1e-7 * max(1.0, abs(polyval(cn, 0.0, d)))
    for a in range(3):     # at rest (v = a = jerk = 0) at both ends
        for d in (1, 2, 3):
            assert abs(polyval(P[0, 1 + 8 * a:9 + 8 * a], 0.0, d)) < 1e-9
            assert abs(polyval(P[-1, 1 + 8 * a:9 + 8 * a], T[-1], d)) < 1e-6


def test_snap_cost_is_minimal_among_feasible_trajectories():
    """Perturbing the interior derivatives (v, a, jerk at a waypoint) keeps every constraint and must not lower the cost:
    rebuild both neighbouring segments from perturbed end conditions and compare."""
    rng = np.random.default_rng(5)
    wp = np.vstack([[0, 0, 3.0], rng.uniform(-5, 5, (3, 3)) + [0, 0, 7.5]])
    T = np.array([1.3, 1.7, 1.1])
    P = tr.minsnap_solve(wp, T)
    J0 = snap_cost(P)

    def rebuild(P, vertex, delta):            # delta [3 axes, 3 derivatives] added at interior `vertex`
        Q = P.copy()
        for a in range(3):
            for s, at_end in ((vertex - 1, True), (vertex, False)):
                c = Q[s, 1 + 8 * a:9 + 8 * a]
                Ts = T[s]
                d = np.array([polyval(c, 0.0, r) for r in range(4)] + [polyval(c, Ts, r) for r in range(4)])
                d[(4 if at_end else 0) + 1:(4 if at_end else 0) + 4] += delta[a]
                A = np.zeros((8, 8))
                for r in range(4):
                    for i in range(r, 8):
                        f = np.prod([i - k for k in range(r)]) if r else 1.0
                        A[r, i] = f * 0.0 ** (i - r) if i > r else f
                        A[4 + r, i] = f * Ts ** (i - r)
                Q[s, 1 + 8 * a:9 + 8 * a] = np.linalg.solve(A, d)
        return Q

    for vertex in (1, 2):
        for _ in range(6):
            delta = rng.normal(0, 0.3, (3, 3))
            assert snap_cost(rebuild(P, vertex, delta)) > J0 * (1 + 1e-9)
            assert snap_cost(rebuild(P, vertex, 1e-3 * delta)) >= J0 * (1 - 1e-9)


@pytest.mark.parametrize("v_max,a_max", [(12.0, 12.0), (10.0, 10.0), (15.0, 5.0), (3.0, 20.0)])
def test_generate_meets_limits_with_one_active(v_max, a_max):
    for index in range(4):
        wp = tr.random_waypoints(7, index)
        P = tr.minsnap_pieces(wp, v_max, a_max)
        x, t = tr.sample_polynomial_trajectory_fast(P, 0.002)
        v = np.linalg.norm(x[:, 7:10], axis=1).max()
        acc = np.linalg.norm(np.diff(x[:, 7:10], axis=0) / 0.002, axis=1).max()
        assert v <= v_max * (1 + 1e-3) and acc <= a_max * (1 + 2e-2)
        assert max(v / v_max, acc / a_max) > 0.97                      # the binding limit is reached, not just respected
        assert np.abs(x[0, :3] - wp[0]).max() < 1e-6 and np.abs(x[-1, :3] - wp[-1]).max() < 0.05


def test_csv_roundtrip_in_reference_format(tmp_path):
    wp = tr.random_waypoints(3, 1)
    P = tr.minsnap_pieces(wp, 12.0, 12.0)
    path = tmp_path / "polynomial_representation.csv"
    tr.write_polynomial_csv(path, P)
    # uav_trajectory.Trajectory.loadcsv: np.loadtxt(filename, delimiter=",", skiprows=1, usecols=range(33), ndmin=2)
    back = np.loadtxt(path, delimiter=",", skiprows=1, usecols=range(33), ndmin=2)
    assert back.shape == P.shape and np.abs(back - P).max() <= 0.5e-6
    assert open(path).readline().startswith("# duration,x^0,x^1")
    xa, ta = tr.sample_polynomial_trajectory_fast(back, 0.01)
    xb, tb = tr.sample_polynomial_trajectory(back, 0.01)
    xc, tc = tr.sample_polynomial_trajectory_native(back, 0.01)
    assert np.array_equal(xa, xb) and np.array_equal(xa, xc) and np.array_equal(ta, tc)     # the vectorised and the C++ sampler are the reference-pinned one


def test_fast_sampler_matches_reference_generated_vectors():
    g = np.load(os.path.join(ROOT, "tests", "golden", "poly_vectors.npz"))
    for c in range(int(g["ncases"])):
        for sampler in (tr.sample_polynomial_trajectory_fast, tr.sample_polynomial_trajectory_native):
            x, t = sampler(g[f"p{c}_pieces"], float(g[f"p{c}_dt"]))
            assert x.shape == g[f"p{c}_x"].shape
            assert np.abs(x - g[f"p{c}_x"]).max() <= 1.5e-6 and np.abs(t - g[f"p{c}_t"]).max() < 1e-9


def test_swarm_is_partition_invariant_and_tracks_estimate():
    a, la = tr.swarm_trajectories(5, 0, 4, kind="minsnap")
    b, lb = tr.swarm_trajectories(5, 2, 2, kind="minsnap")
    assert np.array_equal(la[2:], lb) and np.array_equal(a[2, :lb[0]], b[0, :lb[0]])
    T = tr.minsnap_estimate_times(tr.random_waypoints(5, 0), 12.0, 12.0)
    assert T.shape == (3,) and (T > 0).all()


def jerk_cost(pieces):
    J = 0.0
    for row in pieces:
        T = row[0]
        for a in range(3):
            c = row[1 + 8 * a:9 + 8 * a]
            s = np.array([i * (i - 1) * (i - 2) * c[i] for i in range(3, 8)])     # jerk = sum s_k t^k, k = i - 3
            for i in range(5):
                for j in range(5):
                    J += s[i] * s[j] * T ** (i + j + 1) / (i + j + 1)
    return J


def test_published_linear_stage_of_the_reference_generator():
    """The reference's genTrajectory = mav_trajectory_generation::PolynomialOptimizationNonLinear<8> with derivative_to_optimize = JERK
    (DESIGN.md section 6.1).  Its linear stage is published and reproduced here: estimateSegmentTimes = estimateSegmentTimesNfabian with
    the constant 6.5, t = 2 d / v_max (1 + 6.5 v_max / a_max exp(-2 d / v_max)), then the jerk-optimal solveLinear with the same vertex
    constraints as the snap solve -- and the jerk solve really minimises the jerk cost (it beats the snap solve on it and vice versa)."""
    wp = np.array([[0, 0, 3.0], [5, 0, 6], [5, 5, 9], [-5, 5, 12]])
    v_max, a_max = 10.0, 10.0
    T = tr.minsnap_estimate_times(wp, v_max, a_max)
    d = np.linalg.norm(np.diff(wp, axis=0), axis=1)
    assert np.allclose(T, 2 * d / v_max * (1 + 6.5 * v_max / a_max * np.exp(-2 * d / v_max)), rtol=1e-14)
    Pj, Ps = tr.minsnap_solve_order(wp, T, 3), tr.minsnap_solve_order(wp, T, 4)
    assert np.array_equal(Ps, tr.minsnap_solve(wp, T))
    assert np.array_equal(tr.reference_linear_stage(wp, v_max, a_max, 3), Pj)
    assert jerk_cost(Pj) < jerk_cost(Ps) * (1 - 1e-3) and snap_cost(Ps) < snap_cost(Pj) * (1 - 1e-3)
    for P in (Pj, Ps):      # same constraints either way: through the waypoints, at rest at both ends, v / a / jerk continuous
        for s in range(3):
            for a in range(3):
                c = P[s, 1 + 8 * a:9 + 8 * a]
                assert abs(polyval(c, 0.0) - wp[s, a]) < 1e-9 and abs(polyval(c, T[s]) - wp[s + 1, a]) < 1e-7
                if s < 2:
                    cn = P[s + 1, 1 + 8 * a:9 + 8 * a]
                    for der in (1, 2, 3):
                        assert abs(polyval(c, T[s], der) - polyval(cn, 0.0, der)) < 1e-7 * max(1.0, abs(polyval(cn, 0.0, der)))


# src/trajectory_generation/waypoints/user_defined_waypoints.csv of the reference (data: the `--trajectory 0` flights of the logs)
STATIC_WAYPOINTS = np.array([[0, 0, 3.0], [5, 0, 6], [5, 5, 9], [-5, 5, 12], [-5, -5, 9], [5, -5, 6], [0, 0, 3]])


def test_generators_against_the_logged_references_of_the_static_waypoint_file():
    """What the reference's binary produced for its static waypoint file is in the logs (x_ref of the traj0 runs: v_max = a_max = 10 and
    15 / 5 in the python simulation, 12 / 12 in gazebo).  Duration and peak speed / acceleration of (i) the published linear stage
    (Nfabian times + jerk solve), (ii) this repository's generators (jerk or snap cost on the Nfabian proportions, scaled onto the
    limits) against the logged trajectories -- the gap f3 is documented with, per case.  The binary's nonlinear stage (nlopt Subplex,
    early-stopped) roughly halves the Nfabian times; the scaled generators land within 7 % (jerk) / 10 % (snap) of its durations."""
    rows = []
    for (v_max, a_max), name, dt in (((10.0, 10.0), "log_traj0_v10_a10_gp2.npz", 0.1), ((15.0, 5.0), "log_traj0_v15_a5_gp2.npz", 0.1),
                                     ((12.0, 12.0), "log_gazebo_traj0_v12_a12_gp0.npz", 0.01)):
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        xr = g["x_ref"][int(g["junction"]):] if "junction" in g.files else g["x_ref"]
        assert np.abs(xr[0, :3] - STATIC_WAYPOINTS[0]).max() < 0.05 and np.abs(xr[-1, :3] - STATIC_WAYPOINTS[-1]).max() < 0.05
        logged = (len(xr) * dt, np.linalg.norm(xr[:, 7:10], axis=1).max(), np.linalg.norm(np.diff(xr[:, 7:10], axis=0), axis=1).max() / dt)

        def stats(P):
            x = tr.sample_polynomial_trajectory_native(P, 0.01)[0]
            return P[:, 0].sum(), np.linalg.norm(x[:, 7:10], axis=1).max(), np.linalg.norm(np.diff(x[:, 7:10], axis=0), axis=1).max() / 0.01
        lin = stats(tr.reference_linear_stage(STATIC_WAYPOINTS, v_max, a_max, 3))
        jerk = stats(tr.minsnap_pieces_order(STATIC_WAYPOINTS, v_max, a_max, 3))
        snap = stats(tr.minsnap_pieces(STATIC_WAYPOINTS, v_max, a_max))
        rows.append((v_max, a_max, logged, lin, jerk, snap))
        assert 1.7 < lin[0] / logged[0] < 3.1            # the published first stage is the slow starting point of the binary's optimisation
        assert abs(jerk[0] / logged[0] - 1) < 0.07 and abs(snap[0] / logged[0] - 1) < 0.10
        assert jerk[1] <= v_max * 1.001 and jerk[2] <= a_max * 1.02 and snap[2] <= a_max * 1.02      # ours respect the limits the binary only penalises
    print("\nv_max a_max | logged T vmax amax | linear stage (jerk) | jerk, scaled | snap, scaled (bench workload)")
    for v_max, a_max, *cols in rows:
        print(f"{v_max:5.0f} {a_max:5.0f} | " + " | ".join(f"{T:6.2f} s {v:5.2f} {a:5.2f}" for T, v, a in cols))


def _logged_reference(name, dt):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    xr = g["x_ref"][int(g["junction"]):] if "junction" in g.files else g["x_ref"]
    # the node logs the FIRST ROW OF THE CHUNK (src/mpc_controller_node.py:354-357): with fewer than `skip` rows left that is the
    # trajectory's last row, repeated (src/utils/utils.py:924-929) -- the gazebo log ends with 19 such copies, which are not samples at t_k
    n = len(xr)
    while n > 1 and np.array_equal(xr[n - 2], xr[-1]):
        n -= 1
    if n < len(xr):
        n -= 1
    xr = xr[:n]
    return xr[:, :3], xr[:, 7:10], dt * np.arange(len(xr))


def _piece_fit(t, p, v, i0, i1):
    """Least-squares 7th-order polynomial per axis on the position and velocity samples i0 .. i1-1 (time centred on the piece): residuals."""
    tl = t[i0:i1] - 0.5 * (t[i0] + t[i1 - 1])
    k = np.arange(8)
    A = np.vstack([tl[:, None] ** k, k * tl[:, None] ** np.maximum(k - 1, 0)])
    r = [A @ np.linalg.lstsq(A, np.concatenate([p[i0:i1, a], v[i0:i1, a]]), rcond=None)[0] - np.concatenate([p[i0:i1, a], v[i0:i1, a]]) for a in range(3)]
    return np.concatenate(r)


def _piece_boundaries(t, p, v, wp):
    """First sample of every piece: closest approach to each interior waypoint in turn, then descent on the total residual of the free fits."""
    idx, k0 = [0], 0
    for w in wp[1:-1]:
        k0 += int(np.argmin(np.linalg.norm(p[k0:] - w, axis=1)))
        idx.append(k0)
    idx.append(len(t))
    total = lambda ix: sum(np.sum(_piece_fit(t, p, v, ix[s], ix[s + 1]) ** 2) for s in range(len(wp) - 1))
    best, moved = total(idx), True
    while moved:
        moved = False
        for b in range(1, len(wp) - 1):
            for d in (-40, -10, -3, -1, 1, 3, 10, 40):
                cand = list(idx); cand[b] += d
                if not (cand[b - 1] + 9 <= cand[b] <= cand[b + 1] - 9):
                    continue
                c = total(cand)
                if c < best * (1 - 1e-9):
                    best, idx, moved = c, cand, True
    return idx


def _sample_exact(P, dt, n):
    """Positions and velocities of pieces at t_k = k dt, k < n, without the CSV rounding (the fit below differentiates through it)."""
    ends = np.cumsum(P[:, 0])
    t = dt * np.arange(n)
    s = np.minimum(np.searchsorted(ends, t, side="right"), len(P) - 1)
    tl = t - np.where(s > 0, ends[s - 1], 0.0)
    k = np.arange(8)
    out = np.zeros((n, 6))
    for a in range(3):
        c = P[s, 1 + 8 * a:9 + 8 * a]
        out[:, a] = np.sum(c * tl[:, None] ** k, axis=1)
        out[:, 3 + a] = np.sum(c * k * tl[:, None] ** np.maximum(k - 1, 0), axis=1)
    return out


def test_logged_references_are_points_of_the_generators_family():
    """f3, pinned as far as the reference's data allow (round-5 verdict: recover the binary's pieces from the logs instead of re-implementing
    its optimiser).  The three logged references through waypoints/user_defined_waypoints.csv (x_ref of the traj0 runs):
      1. are chains of six 7th-order pieces: a free least-squares polynomial per piece leaves the 6-decimal rounding of the samples;
      2. are points of the family mpcq_minsnap_from_derivatives spans -- pieces through the waypoints, at rest at both ends, C^3 at the
         interior waypoints, parametrised by the segment times T and the free vertex derivatives d_P: fitted in (T, d_P) it reproduces them
         to what the "%.6f" coefficients of the binary's polynomial CSV allow (5e-7 x T^7: 1e-4 .. 1e-3), with the sampler of this package;
      3. are NOT the linear stage at their own times: their d_P differ from the jerk-optimal ones for the same T by 20 - 50 % of the velocity
         scale and their jerk cost is higher -- the binary's nonlinear stage moved times and free derivatives (an early-stopped run over both:
         DESIGN.md section 6.1), which is why segment times alone (round 5) left 0.6 - 1.2 m.
    What stays unpinned is therefore the binary's optimiser path, not the trajectory family, the piece format or the sampler."""
    from scipy.optimize import least_squares
    rows = []
    for name, dt in (("log_traj0_v10_a10_gp2.npz", 0.1), ("log_traj0_v15_a5_gp2.npz", 0.1), ("log_gazebo_traj0_v12_a12_gp0.npz", 0.01)):
        p, v, t = _logged_reference(name, dt)
        n = len(t)
        idx = _piece_boundaries(t, p, v, STATIC_WAYPOINTS)
        free = np.concatenate([_piece_fit(t, p, v, idx[s], idx[s + 1]) for s in range(6)])
        assert np.sqrt(np.mean(free ** 2)) < 4e-7 and np.abs(free).max() < 2e-6, (name, idx)      # 1. (rounding to 6 decimals: 2.9e-7 rms)
        # 2. fit in (T, d_P); start: boundaries -> times, the linear stage's own derivatives
        T0 = np.diff(np.array([t[i] if i < n else t[-1] + dt for i in idx]))

        def d_of(P):
            return np.array([[[P[s, 2 + 8 * a], 2 * P[s, 3 + 8 * a], 6 * P[s, 4 + 8 * a]] for a in range(3)] for s in range(1, 6)])

        def res(z):
            if np.any(z[:6] < 0.2):
                return np.full(6 * n, 1e3)
            x = _sample_exact(tr.minsnap_from_derivatives(STATIC_WAYPOINTS, z[:6], z[6:].reshape(5, 3, 3), 3)[0], dt, n)
            return np.concatenate([(x[:, :3] - p).ravel(), (x[:, 3:] - v).ravel()])
        z0 = np.concatenate([T0, d_of(tr.minsnap_solve_order(STATIC_WAYPOINTS, T0, 3)).ravel()])
        sol = least_squares(res, z0, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=3000)
        T, d = sol.x[:6], sol.x[6:].reshape(5, 3, 3)
        P, J = tr.minsnap_from_derivatives(STATIC_WAYPOINTS, T, d, 3)
        x = tr.sample_polynomial_trajectory_native(P, dt)[0]           # this package's sampler (save_evals_csv + load_trajectory)
        assert 0 <= len(x) - n <= 25     # the rows np.arange(0, total, dt) gives for the fitted duration (gazebo: the logged rows stop `skip` short of the end)
        m = min(len(x), n)
        e_pos, e_vel = np.abs(x[:m, :3] - p[:m]).max(), np.abs(x[:m, 7:10] - v[:m]).max()
        bound = 5e-7 * np.sum(T.max() ** np.arange(8))                 # what rounding eight coefficients to 6 decimals can move a position by
        assert e_pos < 3 * bound and e_vel < 12 * bound, (name, e_pos, e_vel, bound)
        assert np.sqrt(np.mean(sol.fun ** 2)) < bound
        # 3. not the linear stage at these times
        Pl = tr.minsnap_solve_order(STATIC_WAYPOINTS, T, 3)
        dl = d_of(Pl)
        Jl = tr.minsnap_from_derivatives(STATIC_WAYPOINTS, T, dl, 3)[1]
        xl = _sample_exact(Pl, dt, n)
        off_v = np.abs(d[:, :, 0] - dl[:, :, 0]).max() / np.abs(d[:, :, 0]).max()
        assert J > 1.05 * Jl and off_v > 0.15 and np.abs(xl[:, :3] - p).max() > 0.3
        rows.append((name, T, np.sqrt(np.mean(free ** 2)), e_pos, e_vel, bound, off_v, J / Jl, np.abs(xl[:, :3] - p).max()))
    print("\nlog | fitted segment times | free fit rms | family fit: max pos / vel error (rounding bound) | d_P off the linear optimum | jerk cost ratio | linear stage at T: max pos error")
    for name, T, fr, ep, evl, bd, off, jr, lin in rows:
        print(f"{name[4:-4]:28s} | {np.round(T, 3)} | {fr:.1e} | {ep:.1e} / {evl:.1e} ({bd:.1e}) | {100 * off:.0f} % | {jr:.2f} | {lin:.2f} m")
