"""CPU-only regression of the *product kernel sources* through the test-only lane emulator
(tests/wave_emu: the unchanged mpc_quad_ros_amd/csrc files compiled for the host, one fiber per
lane).  It checks lane logic / LDS layout / barriers of the kernels against the oracle where no
GPU exists; the real parity gate is tests/test_gpu_parity.py on the MI355X."""
import dataclasses
import os
import subprocess

import numpy as np
import pytest

import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wave_emu")
EMU = os.path.join(EMU_DIR, "libmpcq_emu.so")


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)


def make(cfg):
    return Engine(cfg, lib_path=EMU)


def test_emu_teacher_forced_nominal():
    assert pc.case_teacher_forced_log(make, "log_traj1_v10_a10_gp0.npz", 6) < 1e-9


def test_emu_teacher_forced_rgp_active_bounds():
    assert pc.case_teacher_forced_log(make, "log_trajectory_v15_a5_gp2.npz", 10) < 1e-8


def test_emu_free_running_cold_start():
    pc.case_free_running_log(make, "log_traj0_v10_a10_gp2.npz", 8)


def test_emu_explicit_api():
    pc.case_explicit_api(make, B=2, N=5, nb=10)
    pc.case_explicit_api(make, B=2, N=5, nb=0)


def test_emu_swarm_closed_loop_hummingbird():
    assert pc.case_swarm_closed_loop(make, B=3, N=10, nb=10, K=6) < 1e-7


def test_emu_saturating_references_many_working_sets():
    worst, hist, failed = pc.case_saturating_references(make, B=2, K=17)
    assert failed == 0
    print("saturating references: worst", worst, "passes", dict(sorted(hist.items())))
    assert worst < 1e-7
    from mpc_quad_ros_amd.engine import WARM_SKIPPED, qp_fallback, qp_flip, qp_warm_exit
    assert any(qp_fallback(v) and not qp_flip(v) for v in hist) and any(qp_flip(v) for v in hist)
    assert any(qp_warm_exit(v) == WARM_SKIPPED for v in hist)   # fallbacks AND the direct interior-point solves of flipping quadrotors


def test_emu_float_interior_point_breakdown_recovers():
    """fp64 instances run the interior-point iterations of a fallback solve in float (ipm_float_stage); one that breaks down is followed
    by the double interior point from its start.  A build in which EVERY float interior point breaks down (tests/wave_emu `brk`: negative
    hand-over tolerance, so it iterates into an indefinite stage Hessian or its iteration cap)
    must end on the same optimum, with the work word saying so (mpcq_get_qp_work bit 15); the product build reports none here."""
    from mpc_quad_ros_amd.engine import qp_fallback
    subprocess.check_call(["make", "-C", EMU_DIR, "brk"], stdout=subprocess.DEVNULL)
    brk = os.path.join(EMU_DIR, "libmpcq_emu_brk.so")
    seen = {}
    for name, lib in (("product", EMU), ("brk", brk)):
        flags = []
        def make_l(cfg, lib=lib, flags=flags):
            e = Engine(cfg, lib_path=lib)
            step = e.step
            def step_and_record(x):
                out = step(x)
                flags.append((qp_fallback(e.get_qp_iter()).copy(), e.get_qp_float_breakdown().copy()))
                return out
            e.step = step_and_record
            return e
        worst, hist, failed = pc.case_saturating_references(make_l, B=2, K=7)
        fb = np.array([f for f, _ in flags]); bd = np.array([b for _, b in flags])
        assert failed == 0 and worst < 1e-7, (name, worst, failed)
        assert fb.sum() >= 2                                   # the cold start and the saturated periods go through the interior point
        seen[name] = (int(fb.sum()), int(bd.sum()))
        if name == "product":
            assert not bd.any()
        else:
            assert bd[fb].all() and bd[0].all()                # every fallback solve of this build, and the cold start, took the recovery path
    print("fallback solves / breakdowns:", seen)
    # a float interior point that ends at its ITERATION CAP (12 here: it never reaches its negative tolerance and is still healthy at 12)
    # leaves the double one its own budget (it used to find the common counter spent and report MPCQ_SOLVE_MAXITER)
    make_c = lambda cfg: Engine(dataclasses.replace(cfg, qp_max_iter=12), lib_path=brk)
    worst, hist, failed = pc.case_saturating_references(make_c, B=2, K=5)
    # (the cold start of the two quadrotors needs more than 12 double iterations and reports the cap: expected, and counted)
    assert failed <= 2 and worst < 1e-7, (worst, failed, hist)
    assert sum(n for v, n in hist.items() if qp_fallback(v) and v % 1000 > 12) >= 4, hist   # 12 float iterations + the double solve behind them


def test_emu_saturating_references_long_warm_attempts():
    """The same references with the early exits of the warm active-set attempt switched off: many-pass attempts (pins and
    releases over several factorisations) end on the same optimum."""
    tune = dict(abort_pins=-1, abort_wrong=-1, flip_max=-1, warm_max=14, warm_retry=14)
    make_t = lambda cfg: make(dataclasses.replace(cfg, tune=tune))
    worst, hist, failed = pc.case_saturating_references(make_t, B=2, K=11)
    print("saturating references, long warm attempts: worst", worst, "passes", dict(sorted(hist.items())))
    assert failed == 0 and worst < 1e-7
    assert sum(n for v, n in hist.items() if 3 <= v % 1000 <= 14 and v < 1000) >= 4


def test_emu_saturating_references_any_shape_instance():
    """The same references on the any-shape instance (tune.generic_kernel): its interior point keeps the per-input quantities in LDS
    (ipm_run), the shape-specialised one of the tests above in registers (ipm_run_regs) -- both against the oracle, with fallback
    solves in the run."""
    make_g = lambda cfg: make(dataclasses.replace(cfg, tune=dict(generic_kernel=1)))
    worst, hist, failed = pc.case_saturating_references(make_g, B=2, K=9)
    print("saturating references, any-shape instance: worst", worst, "passes", dict(sorted(hist.items())))
    assert failed == 0 and worst < 1e-7
    from mpc_quad_ros_amd.engine import qp_fallback
    assert sum(n for v, n in hist.items() if qp_fallback(v)) >= 2


def test_emu_lane_order_independent(monkeypatch):
    # a missing barrier would make results depend on the order lanes run within a phase
    import ctypes, shutil, tempfile
    w_fwd = pc.case_teacher_forced_log(make, "log_traj0_v15_a5_gp2.npz", 3)
    env = dict(os.environ, MPCQ_EMU_REVERSE="1")
    code = ("import sys; sys.path[:0]=[%r,%r]; import parity_cases as pc, test_emu_parity as t; "
            "print(pc.case_teacher_forced_log(t.make, 'log_traj0_v15_a5_gp2.npz', 3))") % (
        os.path.dirname(EMU_DIR), os.path.dirname(os.path.dirname(EMU_DIR)))
    out = subprocess.check_output(["python", "-c", code], env=env).decode().strip().splitlines()[-1]
    assert float(out) == w_fwd


def _closed_loop_digest():
    """Lockstep launches with the plant at their head (MODE_PLANT_FIRST) and the free-running launch, as a number."""
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    B, N, nb = 2, 10, 10
    traj, lens = swarm_trajectories(3, 0, B)
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    out = []
    for mode in ("sim_steps", "sim_run"):
        e = make(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb)))
        e.set_trajectories(traj, lens); e.sim_reset(x0)
        getattr(e, mode)(4, 2, 5e-3)
        x, w = e.sim_get_state()
        out.append(float(np.sum(x * np.arange(1, 14)) + np.sum(w)))
    return out


def test_emu_lane_order_independent_closed_loop():
    # the plant -> measurement hand-over inside a launch (lane 0 writes, lanes 0..12 read) must not depend on lane order
    fwd = _closed_loop_digest()
    assert fwd[0] == fwd[1]
    env = dict(os.environ, MPCQ_EMU_REVERSE="1")
    code = ("import sys; sys.path[:0]=[%r,%r]; import test_emu_parity as t; print(repr(t._closed_loop_digest()))") % (
        os.path.dirname(EMU_DIR), os.path.dirname(os.path.dirname(EMU_DIR)))
    out = subprocess.check_output(["python", "-c", code], env=env).decode().strip().splitlines()[-1]
    assert eval(out) == fwd


def test_emu_compact_layout_against_oracle():
    """The compact layout of large batches (lds_layout gab = 2: Riccati gains in the per-instance global record, fetched ahead like
    the stage records; r0 / lb / ub written behind the shooting into the space of its records): warm active-set solves,
    interior-point fallbacks and flip-marked quadrotors against the oracle."""
    make_c = lambda cfg: make(dataclasses.replace(cfg, tune=dict(stage_mem="compact")))
    assert pc.case_teacher_forced_log(make_c, "log_trajectory_v15_a5_gp2.npz", 5) < 1e-8
    worst, hist, failed = pc.case_saturating_references(make_c, B=2, K=8)
    assert failed == 0 and worst < 1e-7
    from mpc_quad_ros_amd.engine import qp_fallback
    assert any(qp_fallback(v) for v in hist)


def test_emu_block_order_is_cost_sorted_and_changes_nothing():
    """mpcq::order_kernel (launch order of a lockstep period, large batches): a permutation inside the classes p mod 8, every
    class in ascending cost bin of the previous period's qp_iter (bin 0 = predicted most expensive), and the results of the
    launch do not depend on it (B = 21: classes of unequal size)."""
    from mpc_quad_ros_amd.engine import order_bin
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    B, N = 21, 5
    traj, lens = swarm_trajectories(5, 0, B)
    rng = np.random.default_rng(1)
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    x0[:, :3] += rng.normal(0, 0.5, (B, 3)); x0[:, 7:10] += rng.normal(0, 1.0, (B, 3))
    e1 = make(EngineConfig(batch=B, N=N, quad=hummingbird(), tune=dict(block_order=1)))
    e2 = make(EngineConfig(batch=B, N=N, quad=hummingbird(), tune=dict(block_order=2)))
    for e in (e1, e2):
        e.set_trajectories(traj, lens); e.sim_reset(x0); e.sim_steps(3, 2, 5e-3)
    (xa, wa), (xb, wb) = e1.sim_get_state(), e2.sim_get_state()
    assert np.array_equal(xa, xb) and np.array_equal(wa, wb) and np.array_equal(e1.get_state()["X"], e2.get_state()["X"])
    assert np.array_equal(e1.get_block_order(), np.arange(B))
    # a previous-period record that spans several cost bins, marks included (any values are a valid input)
    it_prev = np.array([0, 1, 2, 1003, 11005, 3, 1, 612006, 1, 2, 1, 9, 1, 1, 14, 1, 2, 1, 1001, 1, 4], np.int32)
    e2.set_solver_state(qp_iter=it_prev)
    e2.sim_steps(1, 2, 5e-3)
    order = e2.get_block_order()
    assert sorted(order) == list(range(B))
    for x in range(8):
        cls = order[x::8]
        assert np.all(cls % 8 == x)
        bins = order_bin(it_prev[cls])
        assert np.all(np.diff(bins) >= 0), (x, cls, bins)
        for k in np.unique(bins):            # stable inside a bin
            assert np.all(np.diff(cls[bins == k]) > 0)
    assert (e2.get_status() == 0).all()
    e1.close(); e2.close()


def test_emu_grouped_sim_steps_change_nothing():
    """mpcq_tuning.groups: mpcq_sim_steps runs the batch as contiguous groups, each in lockstep on a stream of its own (launches of
    [b0, b0 + n) with DevState::b0, the launch order sorted per group with global indices).  Same results as one launch over the batch,
    bit for bit, with and without the cost-sorted order; the groups are unequal (B = 40: 24 + 16)."""
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    B, N = 40, 5
    traj, lens = swarm_trajectories(5, 0, B)
    rng = np.random.default_rng(2)
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    x0[:, :3] += rng.normal(0, 0.5, (B, 3)); x0[:, 7:10] += rng.normal(0, 1.0, (B, 3))
    outs = []
    for tune in (dict(groups=1, block_order=1), dict(groups=2, block_order=2)):
        e = make(EngineConfig(batch=B, N=N, quad=hummingbird(), tune=tune))
        e.set_trajectories(traj, lens); e.sim_reset(x0)
        e.sim_steps(2, 2, 5e-3); e.sim_steps(1, 2, 5e-3)
        assert (e.get_status() == 0).all()
        order = e.get_block_order()
        assert sorted(order) == list(range(B))
        if tune["groups"] == 2 and tune["block_order"] == 2:      # sorted inside each group: a permutation of the group's own indices
            assert sorted(order[:24]) == list(range(24)) and sorted(order[24:]) == list(range(24, 40))
        outs.append((e.sim_get_state(), e.get_state()["X"], e.get_state()["idx"], e.get_tracking_stats()))
        e.close()
    for (xw, X, idx, st) in outs[1:]:
        assert np.array_equal(xw[0], outs[0][0][0]) and np.array_equal(xw[1], outs[0][0][1]) and np.array_equal(X, outs[0][1])
        assert np.array_equal(idx, outs[0][2]) and np.array_equal(st, outs[0][3])


def test_emu_f32_qp_mode_within_budget():
    # TQ = float: state and QP data still formed in double; north_star budget 1e-4 relative control deviation
    assert pc.case_swarm_closed_loop(make, B=2, N=20, nb=10, K=10, precision=1) < 1e-4


def test_emu_f32_saturating_references_through_the_interior_point():
    """The mixed-precision path where a float-only solve is 1e-2 off (round 4): infeasible references, inputs saturated over most of the
    horizon, nearly every solve through the float interior point, the active-set method with fp64 residuals behind it (including the
    last-resort run behind the interior point's final iterations).  Every solve within the 1e-4 budget, none failed, status 0."""
    worst, hist, failed = pc.case_saturating_references(make, B=2, K=14, precision=1)
    print("f32 saturating references: worst", worst, "passes", dict(sorted(hist.items())))
    assert failed == 0 and worst < pc.TOL_TF[1]
    from mpc_quad_ros_amd.engine import qp_fallback
    assert sum(n for v, n in hist.items() if qp_fallback(v)) >= 8


def test_emu_tumbling_flight_is_solved_or_flagged():
    """The f32 validity limit at its edge (tests/test_gpu_parity.py runs periods 100 .. 129 in both precisions): periods 112 .. 121 of the reference's
    tumbling traj2_v10_a10_gp2 flight on the emulator.  Two of them (116, 120) make the working set cycle under the float factorisation: round 5
    returned them 0.13 / 0.43 of full thrust off with status 0; flagged in the first half of round 6; solved since the method falls back to
    single-pin steps once it cycles (MPCQ_MIXED_ONEPIN).  Every solve: status 0 within the budget, or flagged."""
    worst, clean, flagged, worst_flagged = pc.case_tumbling_window(make, 1, first=112, last=122)
    print("emu tumbling window f32:", worst, clean, flagged, worst_flagged)
    assert clean + flagged == 10 and clean >= 9
    worst, clean, flagged, _ = pc.case_tumbling_window(make, 0, first=114, last=121)
    assert flagged == 0 and worst < 1e-8


def test_emu_f32_long_horizon_cold_start_in_flight():
    """N = 50 / nb = 50 in f32, started in flight with a cold iterate (interior-point solves, many pins at once): teacher-forced, every
    solve within the budget.  (The GPU test runs 64 quadrotors x 60 periods of this; here 2 x 3 on the emulator.)"""
    assert pc.case_swarm_closed_loop(make, B=2, N=50, nb=50, K=3, precision=1, start=200) < pc.TOL_TF[1]


def test_emu_facade_mirrors_quad_optimizer(lib=EMU):
    from mpc_quad_ros_amd.quad_opt import quad_optimizer
    from mpc_quad_ros_amd.params import hummingbird, rgp_basis_linspace
    from mpc_quad_ros_amd.host_math import compute_a_drag
    from oracle.oracle import OracleEngine
    B, N, nb = 2, 5, 10
    gpe = dict(basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
    qo = quad_optimizer(hummingbird(), t_horizon=1, n_nodes=N, gpe=gpe, batch=B, lib_path=lib)
    assert qo.optimization_dt == 1 / N and qo.gpe.type == "RGP" and qo.gpe.gp[0].X.shape == (nb,)
    with pytest.raises(ValueError):
        qo.run_optimization(None)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0.5, 0, 0, 0, 0, 0]), (B, 1))
    xr = np.tile(x[:, None, :], (1, N, 1)); xr[:, :, 0] += np.linspace(0, 0.5, N)
    yref, yref_N = qo.set_reference_trajectory(xr)
    assert yref.shape == (B, N, 17) and np.all(yref[:, :, 13:] == 0.16) and np.array_equal(yref_N, xr[:, -1])
    x_opt, w_opt, t_cpu, cost = qo.run_optimization(x)
    assert x_opt.shape == (B, N + 1, 13) and w_opt.shape == (B, N, 4) and cost.shape == (B,)
    o = OracleEngine(qo.cfg); o.set_reference(yref, yref_N); o.solve(x)
    assert np.abs(w_opt[:, 0] - o.get_u(0)).max() < 1e-9
    x_pred = qo.discrete_dynamics(x, w_opt[:, 0], 0.01)
    assert np.abs(x_pred - o.predict_nominal(x, w_opt[:, 0], 0.01)).max() < 1e-13
    with pytest.raises(AssertionError):
        qo.discrete_dynamics(x[0], w_opt[0, 0], 0.01)
    vb, ad = compute_a_drag(x, x_pred, 0.01)
    mu, C = qo.regress_and_update_RGP_model([vb[:, d] for d in range(3)], [ad[:, d] for d in range(3)])
    o.rgp_regress(vb, ad); mu_o, C_o = o.get_rgp()
    assert len(mu) == 3 and mu[0].shape == (B, nb) and C[0].shape == (B, nb, nb)
    assert np.abs(np.stack(mu, 1) - mu_o).max() < 1e-12 and np.abs(np.stack(C, 1) - C_o).max() < 1e-12


def test_emu_lost_quadrotors_are_solved_in_fp64_and_flagged_in_f32():
    """tests/golden/f32_lost_quadrotors.npz on the lane emulator (the GPU test of the same name): the float arithmetic of the emulator breaks
    down on the same two quadrotors (status 1 and 8 where the MI355X reports 8 and 1) and solves the idling one."""
    rows = pc.case_lost_quadrotors(make)
    for row in rows:
        print("%s: fp64 status %d deviation %.1e | f32 status %d deviation %.1e" % row)
    assert [r[1] for r in rows] == [0, 0, 0]

