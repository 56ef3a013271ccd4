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
    assert names == ["mpcq_minsnap_estimate_times", "mpcq_minsnap_generate", "mpcq_minsnap_sample", "mpcq_minsnap_solve", "mpcq_minsnap_write_csv"]
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
