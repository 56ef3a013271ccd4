"""Parity cases shared by the GPU tests (real libmpcq.so) and the lane-emulator tests (same
product sources compiled for the host, tests/wave_emu).  Every case drives the engine under test
and the fp64 CPU oracle with identical inputs through the same C-ABI-shaped surface.

Tolerances (relative control deviation, north_star budget 1e-4):
  F64 device path: 1e-7 teacher-forced (observed <= 3e-10), 1e-6 free-running on contractive windows.
"""
import numpy as np

from helpers import config_for_log, load_golden, random_states
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
from oracle.oracle import OracleEngine

TOL_TF = {0: 1e-7, 1: 1e-4}     # teacher-forced, by precision code (f32: the north_star budget itself, on EVERY solve)
TOL_FREE = {0: 1e-6, 1: 1e-3}
# MPCQ_PRECISION_F32 is a mixed-precision mode since round 5: float factorisation, QP solution refined against residuals evaluated
# in double (mpcq_kernels.hpp: polish_mixed).  Every solve -- warm, cold start, interior-point fallback -- is held to the 1e-4 budget
# and has to report status 0; MPCQ_SOLVE_LOW_ACCURACY (8) would mean the refinement did not converge and fails the tests.


def rel_err(a, b, floor=1e-3):
    """Worst absolute deviation over the whole array relative to the largest reference magnitude (controls live in
    [0, 1], so this is close to an absolute error in units of full thrust)."""
    return np.abs(a - b).max() / max(np.abs(b).max(), floor)


def rel_err_per_instance(a, b, floor=1e-3):
    """Stricter: every instance against its OWN largest reference magnitude; returns the worst instance."""
    a, b = np.asarray(a), np.asarray(b)
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return float((np.abs(a2 - b2).max(axis=1) / np.maximum(np.abs(b2).max(axis=1), floor)).max())


def case_teacher_forced_log(make_engine, name, K, precision=0, check_rgp=True):
    """P1: before every step the engine state is overwritten with the oracle's, so each step
    isolates arithmetic: same (X,U,mu,C,x_pred_prev,idx) in -> compare w, x_pred, cost, mu, C."""
    g = load_golden(name)
    cfg = config_for_log(g, precision=precision)
    e, o = make_engine(cfg), OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None]); o.set_trajectories(g["x_ref"][None])
    worst = 0.0
    for k in range(K):
        e.set_state(**o.get_state())
        w, xp = e.step(g["x_odom"][k][None])
        wo, xpo = o.step(g["x_odom"][k][None])
        status = int(e.get_status()[0])
        assert status == 0, (k, status)
        err = rel_err(w, wo)
        worst = max(worst, err)
        tol_k = TOL_TF[precision]
        assert err < tol_k, (name, k, err, status)
        assert rel_err(xp, xpo, 1.0) < tol_k          # prediction and cost follow the control: same bound as that step's control
        assert abs(e.get_cost()[0] - o.get_cost()[0]) <= (10 if precision == 1 else 1) * tol_k * max(1.0, o.get_cost()[0])
        if cfg.nb and check_rgp:
            mu, C = e.get_rgp(); muo, Co = o.get_rgp()
            assert rel_err(mu, muo, 1.0) < (1e-10 if precision == 0 else 1e-4)
            assert rel_err(C, Co, 1e-2) < (1e-10 if precision == 0 else 1e-4)
        se, so = e.get_state(), o.get_state()
        assert np.array_equal(se["idx"], so["idx"]) and np.array_equal(se["has_prev"], so["has_prev"])
    return worst


def case_tumbling_window(make_engine, precision, first=100, last=130):
    """The reference's own traj2_v10_a10_gp2 flight behind step 100, where its loop loses the quadrotor (cost_solution 0.7 -> 1e4 within thirty
    periods, predictions with |q| far from 1, QP gradient scale 1e6 .. 1e8): teacher-forced against the oracle, which runs the whole flight on the
    logged measurements.  fp64 has to hold its tolerance on every solve.  The mixed-precision mode is outside its validity limit on some of
    these solves (cond x eps32 >= 1: the float factorisation is no contraction): every solve must EITHER hold the 1e-4 budget with status 0 OR
    say so -- MPCQ_SOLVE_LOW_ACCURACY or a failed-solve code -- never a silent miss.  Returns (worst deviation among status-0 solves,
    status-0 solves, flagged solves, worst deviation among flagged ones)."""
    g = load_golden("log_traj2_v10_a10_gp2_whole.npz")
    e, o = make_engine(config_for_log(g, precision=precision)), OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None]); o.set_trajectories(g["x_ref"][None])
    worst, clean, flagged, worst_flagged = 0.0, 0, 0, 0.0
    for k in range(last):
        if k >= first - 3:      # (three periods ahead of the window: the engine's warm-start flags settle into the sequence)
            e.set_state(**o.get_state())
            w, _ = e.step(g["x_odom"][k][None])
        wo, _ = o.step(g["x_odom"][k][None])
        assert int(o.get_status()[0]) == 0, k
        if k < first:
            continue
        status, err = int(e.get_status()[0]), rel_err(w, wo)
        if status == 0:
            assert err < TOL_TF[precision], (k, err)
            worst, clean = max(worst, err), clean + 1
        else:
            assert precision == 1, (k, status)      # fp64 solves everything in this window
            flagged, worst_flagged = flagged + 1, max(worst_flagged, err)
    return worst, clean, flagged, worst_flagged


def case_lost_quadrotors(make_engine):
    """The three solves the widened f32 audit of round 6 turned up on seeds nothing was tuned on (tests/golden/make_lost_quadrotors.py,
    profiles/r6_f32_audit_more.txt): one quadrotor of the bench workload each, its state just before the period.  Two are lost (QP gradient
    scale 1e9, 78 of 80 inputs at a bound), one idles.  fp64 has to solve all three (status 0, the oracle's control to 2e-7 of full thrust:
    these QPs are as ill-conditioned as the workload gets); the mixed-precision mode has to EITHER return the oracle's control to 2e-6 of
    full thrust with status 0 OR say so (MPCQ_SOLVE_LOW_ACCURACY or a failed-solve code) -- never a silent miss.  Returns a list of
    (origin, status fp64, deviation fp64, status f32, deviation f32)."""
    g = load_golden("f32_lost_quadrotors.npz")
    rows = []
    for c in range(int(g["cases"])):
        p = f"c{c}_"
        N, nb = int(g[p + "N"]), int(g[p + "nb"])
        cfg = lambda prec: EngineConfig(batch=1, N=N, T=1.0, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1],
                                        dt_pred=0.01, precision=prec)
        st = {k[len(p) + 3:]: g[k][None] for k in g.files if k.startswith(p + "st_")}
        traj, ln, x = g[p + "traj"][None], np.array([int(g[p + "len"])], np.int32), g[p + "x"][None]
        o = OracleEngine(cfg(0))
        o.set_trajectories(traj, ln); o.set_state(**st)
        wo, _ = o.step(x)
        assert int(o.get_status()[0]) == 0, c
        assert np.abs(wo[0] - g[p + "w64"]).max() < 2e-7, (c, wo[0], g[p + "w64"])      # (what the fp64 engine returned on the GPU box)
        row = [str(g[p + "origin"])]
        for prec in (0, 1):
            e = make_engine(cfg(prec))
            e.set_trajectories(traj, ln); e.set_state(**st)
            e.set_solver_state(qp_iter=np.array([int(g[p + "prev"])], np.int32)); e.sim_reset(x)
            e.sim_steps(1, 2, 5e-3)
            status, dev = int(e.get_status()[0]), float(np.abs(e.sim_get_state()[1][0] - wo[0]).max())
            if prec == 0:
                assert status == 0 and dev < 2e-7, (c, status, dev)
            else:
                assert status != 0 or dev < 2e-6, (c, status, dev)
            row += [status, dev]
            e.close()
        rows.append(tuple(row))
    return rows


def case_f32_every_solve_against_f64(make_engine, B, N, nb, K, seed, preroll=0, dump_prefix=None):
    """EVERY solve of an f32 (mixed-precision) lockstep run of the bench workload against the fp64 engine on the same inputs: before each
    period the fp64 engine's state (iterate, RGP state, cursors, plant state) is overwritten with the f32 engine's, both take the period, the
    controls are compared per quadrotor -- relative to the quadrotor's OWN largest control (floor 1e-2: an idling quadrotor is not judged
    against full thrust).  The fp64 engine matches the CPU oracle to 1e-10 on this workload (test_bench_workload_parity_vs_oracle), so this is
    the oracle check at a sample size the oracle cannot reach (10^5 .. 10^6 solves).  Returns a dict: solves / worst deviation / solves beyond
    1e-4 by reported status, and the list of flagged or out-of-budget solves."""
    import bench
    from mpc_quad_ros_amd.engine import qp_fallback
    refs = bench.workload(seed, 0, B, preroll + K + 10)
    mk = lambda prec: make_engine(EngineConfig(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1],
                                               dt_pred=0.01, precision=prec))
    e32, e64 = mk(1), mk(0)
    for e in (e32, e64):
        e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0, (B, 1)))
    if preroll:
        e32.sim_run(preroll, 2, 5e-3)
    out = {key: dict(solves=0, worst=0.0, beyond=0) for key in ("clean", "flagged", "failed")}
    out.update(fallback_clean_worst=0.0, hits=[], B=B, N=N, nb=nb, seed=seed, periods=K, preroll=preroll)
    saved = 0
    for k in range(K):
        st, prev_it = e32.get_state(), e32.get_qp_iter()
        x = e32.sim_get_state()[0]
        e64.set_state(**st); e64.sim_reset(x)
        e32.sim_steps(1, 2, 5e-3); e64.sim_steps(1, 2, 5e-3)
        w32, w64 = e32.sim_get_state()[1], e64.sim_get_state()[1]
        s32, s64, it = e32.get_status(), e64.get_status(), e32.get_qp_iter()
        assert (s64 == 0).all(), (k, np.flatnonzero(s64))
        dev = np.abs(w32 - w64).max(axis=1) / np.maximum(np.abs(w64).max(axis=1), 1e-2)
        for key, sel in (("clean", s32 == 0), ("flagged", s32 == 8), ("failed", (s32 & 7) != 0)):
            if sel.any():
                o = out[key]
                o["solves"] += int(sel.sum()); o["worst"] = max(o["worst"], float(dev[sel].max())); o["beyond"] += int((dev[sel] > 1e-4).sum())
        fb = qp_fallback(it) & (s32 == 0)
        if fb.any():
            out["fallback_clean_worst"] = max(out["fallback_clean_worst"], float(dev[fb].max()))
        for b in np.flatnonzero((s32 != 0) | (dev > 1e-4)):
            out["hits"].append((preroll + k, int(b), int(s32[b]), int(it[b]), float(dev[b]), float(np.abs(w32[b] - w64[b]).max()), float(np.abs(w64[b]).max())))
            if dump_prefix and saved < 4:      # the state in front of the solve, for a single-quadrotor replay on the lane emulator
                np.savez(f"{dump_prefix}_{saved}.npz", N=N, nb=nb, k=preroll + k, b=b, x=x[b], traj=refs[0][b], len=refs[1][b], prev=prev_it[b], w32=w32[b], w64=w64[b],
                         **{f"st_{name}": v[b] for name, v in st.items()})
                saved += 1
    e32.close(); e64.close()
    return out


def case_free_running_log(make_engine, name, K, precision=0):
    """P2: engine and oracle run independently from the cold start on a contractive window."""
    g = load_golden(name)
    e, o = make_engine(config_for_log(g, precision=precision)), OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None]); o.set_trajectories(g["x_ref"][None])
    worst = 0.0
    for k in range(K):
        w, _ = e.step(g["x_odom"][k][None])
        wo, _ = o.step(g["x_odom"][k][None])
        worst = max(worst, rel_err(w, wo))
    assert worst < TOL_FREE[precision], worst
    # and against the reference's own logged acados outputs
    assert np.abs(w[0] - g["w_odom"][K - 1]).max() < 1e-4
    return worst


def case_explicit_api(make_engine, B=4, N=10, nb=10, precision=0, seed=0):
    """acados-style path: set_reference / set_params / solve / get, predict_nominal, rgp_regress."""
    rng = np.random.default_rng(seed)
    kw = dict(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, dt_pred=0.01)
    if nb:
        kw.update(basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
    e, o = make_engine(EngineConfig(precision=precision, **kw)), OracleEngine(EngineConfig(**kw))
    x0 = random_states(rng, B, 0.3)
    xr = random_states(rng, B, 0.3)
    yref = np.zeros((B, N, 17)); yref[:, :, :13] = xr[:, None, :]; yref[:, :, 13:] = 0.16
    yref[:, :, 0] += np.linspace(0, 1, N)[None, :]
    yrefN = yref[:, -1, :13].copy()
    mu = rng.normal(0, 0.5, (B, 3 * nb))
    for eng in (e, o):
        eng.set_reference(yref, yrefN)
        if nb:
            eng.set_params(mu)
    for it in range(3):
        e.solve(x0); o.solve(x0)
        assert ((e.get_status() & 7) == 0).all()
        for stage in (0, 1, N - 1):
            assert rel_err(e.get_u(stage), o.get_u(stage)) < TOL_TF[precision] * (10 if it else 1)
        for stage in (0, 1, N):
            assert rel_err(e.get_x(stage), o.get_x(stage), 1.0) < TOL_TF[precision] * (10 if it else 1)
        assert np.abs(e.get_x(0) - x0).max() < 1e-6
        assert rel_err(e.get_cost(), o.get_cost(), 1.0) < TOL_TF[precision] * 10
    u = rng.uniform(0, 1, (B, 4))
    assert rel_err(e.predict_nominal(x0, u, 0.01), o.predict_nominal(x0, u, 0.01), 1.0) < (1e-13 if precision == 0 else 1e-5)
    if nb:
        for _ in range(3):
            vb, ad = rng.normal(0, 4, (B, 3)), rng.normal(0, 2, (B, 3))
            e.rgp_regress(vb, ad); o.rgp_regress(vb, ad)
        mu_e, C_e = e.get_rgp(); mu_o, C_o = o.get_rgp()
        assert rel_err(mu_e, mu_o, 1.0) < (1e-11 if precision == 0 else 1e-4)
        assert rel_err(C_e, C_o, 1e-2) < (1e-11 if precision == 0 else 1e-4)


def case_swarm_closed_loop(make_engine, B, N, nb, K, precision=0, seed=1, plant_sub=2, start=0, min_changes=0, teacher=None):
    """Synthetic random-waypoint swarm (the bench workload family), host-driven closed loop with the
    oracle's drag plant; engine and oracle free-running side by side on identical measurements.
    start > 0: the run begins `start` samples into the references, on the reference state, with a cold iterate
    (interior-point solves first, then a fast stretch where inputs saturate); min_changes: quadrotor-steps that must
    have gone through more than one working set.
    teacher (default: on for f32): the engine's state is overwritten with the oracle's before every step, so every solve is
    judged on its own.  Every solve of either precision has to report status 0; returns the worst per-quadrotor deviation."""
    if teacher is None:
        teacher = precision == 1
    kw = dict(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, dt_pred=0.01)
    if nb:
        kw.update(basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
    e, o = make_engine(EngineConfig(precision=precision, **kw)), OracleEngine(EngineConfig(**kw))
    traj, lens = swarm_trajectories(seed, 0, B)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    if start:
        assert lens.min() > start + K, lens.min()
        traj, lens = np.ascontiguousarray(traj[:, start:]), lens - start
        x = traj[:, 0].copy()
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    worst, changes, fallbacks = 0.0, 0, 0
    for k in range(K):
        if teacher:
            e.set_state(**o.get_state())
        w, xp = e.step(x)
        wo, xpo = o.step(x)
        status = e.get_status()
        assert (status == 0).all(), (k, status)
        its = e.get_qp_iter()
        changes += int(((its % 1000) > 1).sum())
        fallbacks += int((((its // 1000) % 10) != 0).sum() + (its % 1000 == 0).sum())
        worst = max(worst, rel_err_per_instance(w, wo, floor=1e-2), rel_err(w, wo))
        for _ in range(plant_sub):
            x = o.plant_update(x, wo, 5e-3)
    se, so = e.get_tracking_stats(), o.get_tracking_stats()
    assert np.allclose(se[:4], so, rtol=1e-6 if precision == 0 else 1e-3, atol=1e-9)      # same measurements on both sides
    assert changes >= min_changes, changes
    if precision == 1:
        print(f"f32 swarm: {B * K} instance-steps, {fallbacks} through the interior point, worst deviation {worst:.2e}")
    return worst


def case_saturating_references(make_engine, B=3, K=40, precision=0):
    """Teacher-forced run on deliberately infeasible references (fast lateral sinusoid + vertical steps): the thrust
    saturates on large parts of the horizon and the working set changes by many inputs per step, so the warm
    active-set attempt goes through many working sets (restarted factorisations, bulk pins / releases) and regularly
    gives up to the interior-point fallback.  Returns (worst relative control deviation over the instance-steps that
    report success, histogram of pass counts, number of instance-steps reporting a failed solve)."""
    N, nb = 20, 10
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb))
    e, o = make_engine(EngineConfig(precision=precision, **kw)), OracleEngine(EngineConfig(**kw))
    T = 60 + K * 5
    traj = np.zeros((B, T, 13)); traj[:, :, 3] = 1.0
    t = np.arange(T) * 0.01
    for b in range(B):
        A, w = 2.0 + b, 3.0 + 0.7 * b
        traj[b, :, 0] = A * np.sin(w * t); traj[b, :, 7] = A * w * np.cos(w * t)
        traj[b, :, 2] = 3.0 + 1.5 * np.sign(np.sin(2.0 * t + b))
    lens = np.full(B, T, dtype=np.int32)
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    worst, hist, failed = 0.0, {}, 0
    for k in range(K):
        w_e, _ = e.step(x)
        w_o, _ = o.step(x)
        ok = e.get_status() == 0                                   # (a failed solve holds the previous control and is counted)
        failed += int(((e.get_status() & 7) != 0).sum())
        assert np.isfinite(w_e).all() and w_e.min() >= 0.0 and w_e.max() <= 1.0     # a failed solve holds the previous control
        for v in e.get_qp_iter():
            hist[int(v)] = hist.get(int(v), 0) + 1
        if ok.any():
            worst = max(worst, rel_err(w_e[ok], w_o[ok]))
        x = o.plant_control_period(x, w_o, 0.01, 5e-3)[0]
        st = o.get_state()
        e.set_state(X=st["X"], U=st["U"], mu=st["mu"], C=st["C"], x_pred_prev=st["x_pred_prev"], has_prev=st["has_prev"], idx=st["idx"])
    return worst, hist, failed
