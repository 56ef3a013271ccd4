"""Seeded synthetic random-waypoint reference trajectories (host side, numpy).

Family: the node's 'random' request — start at the hover point, 3 waypoints uniform in a
+-[5,5,5] m cube centred at [0,0,7.5] (src/trajectory_generator_node.py:159-167,
src/trajectory_generation/TrajectoryGenerator.py:133-163), v_max / a_max limits, sampled at 100 Hz,
q_ref = [1,0,0,0], rates 0 (TrajectoryGenerator.py:223-244).  The reference shells out to a
prebuilt min-snap binary (genTrajectory) that cannot run here; this generator replaces it with a
C2 cubic-spline path through the waypoints traversed with a rest-to-rest quintic time law whose
duration is stretched until max|v| <= v_max and max|a| <= a_max.

Instance i of a swarm draws from Generator(seed, i): the trajectory of a quadrotor depends only on
(seed, global index), never on how the swarm is sharded over ranks.
"""
from __future__ import annotations

import numpy as np

HOVER = np.array([0.0, 0.0, 3.0])          # src/mpc_controller_node.py:119
NX = 13


def _natural_cubic_spline(P):
    """C2 natural cubic spline through P[0..n] at unit knot spacing -> per-segment coefficients."""
    n = P.shape[0] - 1
    A = np.zeros((n + 1, n + 1))
    rhs = np.zeros((n + 1, P.shape[1]))
    A[0, 0] = A[n, n] = 1.0
    for i in range(1, n):
        A[i, i - 1], A[i, i], A[i, i + 1] = 1.0, 4.0, 1.0
        rhs[i] = 6.0 * (P[i + 1] - 2 * P[i] + P[i - 1])
    M = np.linalg.solve(A, rhs)                      # second derivatives at the knots
    a = P[:-1]
    b = (P[1:] - P[:-1]) - (2 * M[:-1] + M[1:]) / 6.0
    c = M[:-1] / 2.0
    d = (M[1:] - M[:-1]) / 6.0
    return a, b, c, d


def _eval_path(coef, s):
    a, b, c, d = coef
    n = a.shape[0]
    k = np.minimum(np.floor(s).astype(int), n - 1)
    t = (s - k)[:, None]
    p = a[k] + t * (b[k] + t * (c[k] + t * d[k]))
    dp = b[k] + t * (2 * c[k] + 3 * t * d[k])
    ddp = 2 * c[k] + 6 * t * d[k]
    return p, dp, ddp


def random_waypoint_trajectory(seed: int, index: int, v_max: float = 12.0, a_max: float = 12.0,
                               dt: float = 0.01, num_waypoints: int = 3, hsize=(5.0, 5.0, 5.0),
                               start=HOVER):
    """Returns x_ref [T, 13] sampled every dt (p, q=[1,0,0,0], v, r=0)."""
    rng = np.random.default_rng([int(seed), int(index)])
    hs = np.asarray(hsize, dtype=float)
    centre = np.array([0.0, 0.0, 1.5 * hs[2]])
    wps = [np.asarray(start, dtype=float)]
    for _ in range(num_waypoints):
        wps.append(rng.uniform(-hs, hs) + centre)
    P = np.array(wps)
    coef = _natural_cubic_spline(P)
    n = P.shape[0] - 1
    # rest-to-rest quintic time law s(tau) = n*(10 tau^3 - 15 tau^4 + 6 tau^5), tau = t/D
    tau = np.linspace(0.0, 1.0, 2001)
    s = n * (10 * tau**3 - 15 * tau**4 + 6 * tau**5)
    ds = n * (30 * tau**2 - 60 * tau**3 + 30 * tau**4)
    dds = n * (60 * tau - 180 * tau**2 + 120 * tau**3)
    p, dp, ddp = _eval_path(coef, s)
    v1 = np.linalg.norm(dp * ds[:, None], axis=1).max()                       # |v| at D = 1
    a1 = np.linalg.norm(ddp * (ds**2)[:, None] + dp * dds[:, None], axis=1).max()
    D = max(v1 / v_max, np.sqrt(a1 / a_max), 1.0)
    ts = np.arange(0.0, D, dt)                                                # as save_evals_csv: arange(0, duration, dt)
    tau = ts / D
    s = n * (10 * tau**3 - 15 * tau**4 + 6 * tau**5)
    ds = n * (30 * tau**2 - 60 * tau**3 + 30 * tau**4) / D
    p, dp, _ = _eval_path(coef, s)
    x = np.zeros((len(ts), NX))
    x[:, 0:3] = p
    x[:, 3] = 1.0
    x[:, 7:10] = dp * ds[:, None]
    return x


def random_waypoints(seed: int, index: int, num_waypoints: int = 3, hsize=(5.0, 5.0, 5.0), start=HOVER):
    """Start point + num_waypoints uniform in the cube (generate_random_waypoints, TrajectoryGenerator.py:133-163);
    the same draws as random_waypoint_trajectory(seed, index)."""
    key = [int(s) for s in np.atleast_1d(seed)] + [int(index)]
    rng = np.random.default_rng(key)
    hs = np.asarray(hsize, dtype=float)
    centre = np.array([0.0, 0.0, 1.5 * hs[2]])
    return np.array([np.asarray(start, dtype=float)] + [rng.uniform(-hs, hs) + centre for _ in range(num_waypoints)])


# ---------------------------------------------------------------------------------------------
# Minimum-snap references: what the reference gets from its prebuilt genTrajectory binary
# (TrajectoryGenerator.sample_trajectory, :177-206), here from libmpcq_traj.so (csrc/minsnap.cpp, include/mpcq_traj.h).
_TRAJ_LIB = None


def _traj_lib():
    global _TRAJ_LIB
    if _TRAJ_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmpcq_traj.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `make -C mpc_quad_ros_amd/csrc`")
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.mpcq_minsnap_estimate_times.argtypes = [dp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, dp]
        lib.mpcq_minsnap_solve.argtypes = [dp, ctypes.c_int32, dp, dp]
        lib.mpcq_minsnap_solve_order.argtypes = [dp, ctypes.c_int32, dp, ctypes.c_int32, dp]
        lib.mpcq_minsnap_linear.argtypes = [dp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32, dp]
        lib.mpcq_minsnap_generate.argtypes = [dp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, dp]
        lib.mpcq_minsnap_generate_order.argtypes = [dp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32, dp]
        lib.mpcq_minsnap_from_derivatives.argtypes = [dp, ctypes.c_int32, dp, dp, ctypes.c_int32, dp, dp]
        lib.mpcq_minsnap_write_csv.argtypes = [ctypes.c_char_p, dp, ctypes.c_int32]
        lib.mpcq_minsnap_sample.argtypes = [dp, ctypes.c_int32, ctypes.c_double, dp, ctypes.c_int32]
        _TRAJ_LIB = lib
    return _TRAJ_LIB


def _dptr(a):
    import ctypes
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def minsnap_estimate_times(waypoints, v_max, a_max):
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    T = np.zeros(len(wp) - 1)
    if _traj_lib().mpcq_minsnap_estimate_times(_dptr(wp), len(wp), float(v_max), float(a_max), _dptr(T)):
        raise ValueError("bad waypoints / limits")
    return T


def minsnap_solve(waypoints, times):
    """Minimum-snap 7th-order pieces [n-1, 33] through the waypoints for given segment times."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    T = np.ascontiguousarray(times, dtype=np.float64).reshape(len(wp) - 1)
    pieces = np.zeros((len(wp) - 1, 33))
    rc = _traj_lib().mpcq_minsnap_solve(_dptr(wp), len(wp), _dptr(T), _dptr(pieces))
    if rc:
        raise ValueError(f"mpcq_minsnap_solve failed ({rc})")
    return pieces


def minsnap_solve_order(waypoints, times, derivative_to_optimize):
    """The linear solve with the cost on derivative 4 (snap), 3 (jerk: the reference's binary) or 2 (acceleration)."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    T = np.ascontiguousarray(times, dtype=np.float64)
    pieces = np.zeros((len(wp) - 1, 33))
    rc = _traj_lib().mpcq_minsnap_solve_order(_dptr(wp), len(wp), _dptr(T), int(derivative_to_optimize), _dptr(pieces))
    if rc:
        raise ValueError(f"mpcq_minsnap_solve_order failed ({rc})")
    return pieces


def minsnap_from_derivatives(waypoints, times, d_free, derivative_to_optimize=3):
    """Pieces [n-1,33] and cost for given segment times and free vertex derivatives d_free [n-2,3 axes,3: v,a,j] (both ends at rest): the
    map the nonlinear stage of the reference's generator evaluates per iterate (include/mpcq_traj.h: mpcq_minsnap_from_derivatives)."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64)
    T = np.ascontiguousarray(times, dtype=np.float64)
    d = np.ascontiguousarray(d_free, dtype=np.float64)
    if wp.ndim != 2 or wp.shape[1] != 3 or T.shape != (len(wp) - 1,) or d.shape != (len(wp) - 2, 3, 3):
        raise ValueError("waypoints [n,3], times [n-1], d_free [n-2,3,3]")
    pieces = np.zeros((len(wp) - 1, 33))
    cost = np.zeros(1)
    rc = _traj_lib().mpcq_minsnap_from_derivatives(_dptr(wp), len(wp), _dptr(T), _dptr(d), int(derivative_to_optimize), _dptr(pieces), _dptr(cost))
    if rc:
        raise ValueError(f"mpcq_minsnap_from_derivatives failed ({rc})")
    return pieces, float(cost[0])


def reference_linear_stage(waypoints, v_max, a_max, derivative_to_optimize=3):
    """The linear stage of the reference's genTrajectory as published (mav_trajectory_generation): estimateSegmentTimes with the
    constant 6.5, then PolynomialOptimization<8>::solveLinear for derivative_to_optimize (3 = jerk).  pieces [n - 1, 33]."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    pieces = np.zeros((len(wp) - 1, 33))
    rc = _traj_lib().mpcq_minsnap_linear(_dptr(wp), len(wp), float(v_max), float(a_max), int(derivative_to_optimize), _dptr(pieces))
    if rc:
        raise ValueError(f"mpcq_minsnap_linear failed ({rc})")
    return pieces


def minsnap_pieces_order(waypoints, v_max, a_max, derivative_to_optimize):
    """minsnap_pieces with the cost on another derivative (3 = the jerk cost of the reference's binary)."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    pieces = np.zeros((len(wp) - 1, 33))
    rc = _traj_lib().mpcq_minsnap_generate_order(_dptr(wp), len(wp), float(v_max), float(a_max), int(derivative_to_optimize), _dptr(pieces))
    if rc:
        raise ValueError(f"mpcq_minsnap_generate_order failed ({rc})")
    return pieces


def minsnap_pieces(waypoints, v_max, a_max):
    """genTrajectory -i waypoints --v_max .. --a_max ..: pieces [n-1, 33] with the times scaled onto the limits."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    pieces = np.zeros((len(wp) - 1, 33))
    rc = _traj_lib().mpcq_minsnap_generate(_dptr(wp), len(wp), float(v_max), float(a_max), _dptr(pieces))
    if rc:
        raise ValueError(f"mpcq_minsnap_generate failed ({rc})")
    return pieces


def write_polynomial_csv(path, pieces):
    pieces = np.ascontiguousarray(pieces, dtype=np.float64)
    if _traj_lib().mpcq_minsnap_write_csv(str(path).encode(), _dptr(pieces), len(pieces)):
        raise OSError(f"cannot write {path}")


def sample_polynomial_trajectory_native(pieces, dt: float = 0.01):
    """sample_polynomial_trajectory in C++ (mpcq_minsnap_sample: same operations in the same order, bit-identical; releases the
    GIL, so many trajectories can be sampled from Python threads)."""
    pieces = np.ascontiguousarray(np.atleast_2d(np.asarray(pieces, dtype=np.float64))[:, :33])
    cap = int(np.ceil(float(np.sum(pieces[:, 0])) / dt)) + 2
    x = np.zeros((cap, NX))
    n = _traj_lib().mpcq_minsnap_sample(_dptr(pieces), len(pieces), float(dt), _dptr(x), cap)
    if n < 0:
        raise ValueError("mpcq_minsnap_sample failed")
    return x[:n], np.round(np.arange(n) * dt, 6)


def minsnap_trajectory(seed: int, index: int, v_max: float = 12.0, a_max: float = 12.0, dt: float = 0.01, **kw):
    """The node's 'random' request through the min-snap generator: x_ref [T, 13] sampled every dt."""
    pieces = minsnap_pieces(random_waypoints(seed, index, **kw), v_max, a_max)
    return sample_polynomial_trajectory_fast(pieces, dt)[0]


def minsnap_mission(seed: int, index: int, min_samples: int, v_max: float = 12.0, a_max: float = 12.0, dt: float = 0.01, **kw):
    """Continuous operation of the node: when a trajectory is finished the next random one is requested from where the
    quadrotor stands (src/mpc_controller_node.py:374-399, request_trajectory(x, type)).  A mission is that chain of min-snap
    flights, each through 3 fresh random waypoints starting at the end point of the previous one, until at least
    min_samples reference rows exist.  Depends only on (seed, index)."""
    rows, start, leg, n = [], np.asarray(kw.pop("start", HOVER), dtype=float), 0, 0
    while n < min_samples:
        wp = random_waypoints([int(seed), 7919 * (leg + 1)], index, start=start, **kw)
        x = sample_polynomial_trajectory_native(minsnap_pieces(wp, v_max, a_max), dt)[0]
        rows.append(x)
        n += len(x)
        start, leg = wp[-1], leg + 1
    return np.concatenate(rows)


def _mission_chunk(args):
    """Worker: missions of a contiguous index block, as float32-free compact rows (positions and velocities; the other
    columns of a reference are constants)."""
    seed, first, count, min_samples, kw = args
    return [np.ascontiguousarray(minsnap_mission(seed, first + i, min_samples, **kw)[:, [0, 1, 2, 7, 8, 9]]) for i in range(count)]


def swarm_missions(seed: int, first_index: int, count: int, min_samples: int, workers=None, max_rows=None, **kw):
    """Padded batch of missions (see minsnap_mission): (traj [count, Tmax, 13], lengths [count]).  Large batches are
    generated by forked worker processes (one contiguous index block each; a mission depends only on (seed, index), so the
    result does not depend on the partition): call this BEFORE the process touches the GPU.
    max_rows: keep only the first max_rows rows of every mission (a run of K periods at horizon N with chunk stride `skip`
    reads rows < K + N skip; the rest of the last flight only costs host memory)."""
    import os
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(workers, 16, count // 64))
    if workers > 1:
        import multiprocessing as mp
        block = (count + 4 * workers - 1) // (4 * workers)
        jobs = [(seed, first_index + lo, min(block, count - lo), min_samples, kw) for lo in range(0, count, block)]
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_mission_chunk, jobs)
        trajs = [t for part in parts for t in part]
    else:
        trajs = _mission_chunk((seed, first_index, count, min_samples, kw))
    if max_rows is not None:
        trajs = [t[:max_rows] for t in trajs]
    lens = np.array([t.shape[0] for t in trajs], dtype=np.int32)
    out = np.zeros((count, int(lens.max()), NX))
    out[:, :, 3] = 1.0
    for i, t in enumerate(trajs):
        n = t.shape[0]
        out[i, :n, 0:3] = t[:, 0:3]; out[i, :n, 7:10] = t[:, 3:6]
        out[i, n:, 0:3] = t[-1, 0:3]; out[i, n:, 7:10] = t[-1, 3:6]
    return out, lens


def swarm_trajectories(seed: int, first_index: int, count: int, kind: str = "spline", **kw):
    """Padded batch for Engine.set_trajectories: (traj [count, Tmax, 13], lengths [count]).
    kind: 'spline' (cubic-spline path with a quintic time law) or 'minsnap' (the reference's trajectory family)."""
    gen = {"spline": random_waypoint_trajectory, "minsnap": minsnap_trajectory}[kind]
    trajs = [gen(seed, first_index + i, **kw) for i in range(count)]
    lens = np.array([t.shape[0] for t in trajs], dtype=np.int32)
    Tmax = int(lens.max())
    out = np.zeros((count, Tmax, NX))
    for i, t in enumerate(trajs):
        out[i, :t.shape[0]] = t
        out[i, t.shape[0]:] = t[-1]
    return out, lens


def shard_range(total: int, rank: int, world: int):
    """Contiguous block partition of `total` instances over `world` ranks."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------
# Closed-form circle references of the reference's TrajectoryGenerator
# (src/trajectory_generation/TrajectoryGenerator.py:41-130, loaded back through load_trajectory :223-244,
# i.e. with the CSV's 6-decimal rounding, q = [1,0,0,0], rates 0).  kind: 'accelerating' (:41-78),
# 'constant' (:82-103), 'acc_dec' (:105-131).
def circle_trajectory(kind: str, radius: float, v_max: float, dt: float = 0.01, t_max: float = 10.0,
                      start_point=(0.0, 0.0, 0.0)):
    sp = np.asarray(start_point, dtype=float)
    w_max = v_max / radius
    if kind == "accelerating":
        ts = np.arange(0, t_max, dt)
        n = len(ts)
        k = ((np.arange(n) + 1) / float(n) * 2) - 1
        w = (np.sin((k * 2 * np.pi + np.pi * 3 / 2) * 0.5) + 1) / 2 * w_max
        phi = np.cumsum(w * dt)
    elif kind == "constant":
        ts = np.arange(0, 2 * np.pi / w_max, dt)
        w = np.full(len(ts), w_max)
        phi = np.cumsum(w * dt)
    elif kind == "acc_dec":
        acc0 = w_max * w_max / 2.0 / np.pi
        t_mid = w_max / acc0
        ts = np.arange(0, 2 * t_mid, dt)
        acc = np.where(ts < t_mid, acc0, -acc0)
        w = np.cumsum(acc * dt)
        phi = np.cumsum(w * dt)
    else:
        raise ValueError("kind must be 'accelerating', 'constant' or 'acc_dec'")
    p = np.stack([radius * np.cos(phi) - radius, radius * np.sin(phi), np.zeros_like(phi)], axis=1) + sp
    v = np.stack([-radius * w * np.sin(phi), radius * w * np.cos(phi), np.zeros_like(phi)], axis=1)
    x = np.zeros((len(ts), NX))
    x[:, 0:3] = np.round(p, 6)      # the reference writes '%.6f' CSV and reads it back
    x[:, 3] = 1.0
    x[:, 7:10] = np.round(v, 6)
    return x, np.round(ts, 6)


# ---------------------------------------------------------------------------------------------
# Piecewise-polynomial references in the format the reference's min-snap generator writes
# (src/trajectory_generation/uav_trajectory.py:116-129: one row per piece = duration, then 8 ascending coefficients for
# x, y, z, yaw), sampled like TrajectoryGenerator.save_evals_csv (:203-215) and read back like load_trajectory (:223-244):
# positions / velocities rounded to the CSV's 6 decimals, q = [1,0,0,0], body rates 0.  The min-snap solve itself is a
# prebuilt binary in the reference (genTrajectory) and is not reproduced.
def sample_polynomial_trajectory(pieces, dt: float = 0.01):
    pieces = np.atleast_2d(np.asarray(pieces, dtype=float))
    if pieces.shape[1] < 33:
        raise ValueError("a piece is [duration, x^0..x^7, y^0..y^7, z^0..z^7, yaw^0..yaw^7]")
    dur = pieces[:, 0]
    total = float(np.sum(dur))
    ts = np.arange(0, total, dt)
    x = np.zeros((len(ts), NX))
    x[:, 3] = 1.0
    starts = np.concatenate(([0.0], np.cumsum(dur)[:-1]))
    for k, t in enumerate(ts):
        # piece lookup with the reference's running sum (uav_trajectory.py:146-150)
        cur, row, tl = 0.0, pieces[-1], t - starts[-1]
        for r in pieces:
            if t < cur + r[0]:
                row, tl = r, t - cur
                break
            cur = cur + r[0]
        for a in range(3):
            c = row[1 + 8 * a:9 + 8 * a]
            p = 0.0
            for i in range(8):                     # Horner, highest power first (uav_trajectory.py:22-28)
                p = p * tl + c[7 - i]
            d = [(i + 1) * c[i + 1] for i in range(7)]
            v = 0.0
            for i in range(7):
                v = v * tl + d[6 - i]
            x[k, a] = p
            x[k, 7 + a] = v
    x[:, 0:3] = np.round(x[:, 0:3], 6)
    x[:, 7:10] = np.round(x[:, 7:10], 6)
    return x, np.round(ts, 6)


def sample_polynomial_trajectory_fast(pieces, dt: float = 0.01):
    """sample_polynomial_trajectory for bulk use: the same piece lookup and Horner order, vectorised over the samples
    (bit-identical results; tests/test_minsnap.py)."""
    pieces = np.atleast_2d(np.asarray(pieces, dtype=float))
    dur = pieces[:, 0]
    ts = np.arange(0, float(np.sum(dur)), dt)
    # running-sum lookup: the first piece with t < cur + duration, cur accumulated left to right
    ends = np.zeros(len(dur))
    cur = 0.0
    for k, d in enumerate(dur):
        ends[k] = cur + d
        cur = cur + d
    k = np.minimum(np.searchsorted(ends, ts, side="right"), len(dur) - 1)
    starts = np.concatenate(([0.0], ends[:-1]))
    tl = ts - starts[k]
    x = np.zeros((len(ts), NX))
    x[:, 3] = 1.0
    for a in range(3):
        c = pieces[k, 1 + 8 * a:9 + 8 * a]                      # [T, 8]
        p = np.zeros(len(ts))
        for i in range(8):
            p = p * tl + c[:, 7 - i]
        v = np.zeros(len(ts))
        for i in range(7):
            v = v * tl + (7 - i) * c[:, 7 - i]
        x[:, a] = p
        x[:, 7 + a] = v
    x[:, 0:3] = np.round(x[:, 0:3], 6)
    x[:, 7:10] = np.round(x[:, 7:10], 6)
    return x, np.round(ts, 6)
