"""Edge cases of the engine surface, run through the lane emulator on CPU (same product sources) and,
with -m gpu, on the MI355X: ragged / exhausted trajectories (hold last row), cursor bookkeeping, reset,
state round trip, argument errors, reference-format logging."""
import os
import subprocess

import numpy as np
import pytest

from helpers import random_states
from mpc_quad_ros_amd import _lib
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.host_math import get_reference_chunk
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from oracle.oracle import OracleEngine, reference_chunk

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wave_emu")
EMU = os.path.join(EMU_DIR, "libmpcq_emu.so")


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)


def engines(lib, **kw):
    return Engine(EngineConfig(**kw), lib_path=lib), OracleEngine(EngineConfig(**kw))


def _ragged_and_exhausted(lib):
    B, N = 3, 5
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=0, dt_pred=0.01, skip=2)
    e, o = engines(lib, **kw)
    rng = np.random.default_rng(4)
    lens = np.array([4, 9, 30], dtype=np.int32)          # shorter than one chunk / one partial chunk / long
    traj = np.zeros((B, 30, 13)); traj[:, :, 3] = 1.0
    traj[:, :, :3] = np.cumsum(rng.normal(0, 0.01, (B, 30, 3)), axis=1) + np.array([0, 0, 3.0])
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    for k in range(12):                                   # runs every cursor past the end of its trajectory
        w, xp = e.step(x); wo, xpo = o.step(x)
        assert np.abs(w - wo).max() < 1e-8 and (e.get_status() == 0).all()
        x = xpo
    se, so = e.get_state(), o.get_state()
    assert np.array_equal(se["idx"], so["idx"]) and np.array_equal(se["idx"], np.full(B, 12))
    # host restatement of the chunk agrees with the reference-generated oracle function too
    for b in range(B):
        for idx in (0, 3, lens[b] - 1, lens[b], lens[b] + 5):
            assert np.array_equal(get_reference_chunk(traj[b, :lens[b]], idx, N, 2), reference_chunk(traj[b, :lens[b]], idx, N, 2))


def _reset_and_state_roundtrip(lib):
    B, N, nb = 2, 5, 10
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb))
    e, o = engines(lib, **kw)
    traj = np.zeros((B, 40, 13)); traj[:, :, 3] = 1.0; traj[:, :, 2] = 3.0; traj[:, :, 0] = np.linspace(0, 2, 40)
    e.set_trajectories(traj); o.set_trajectories(traj)
    x = random_states(np.random.default_rng(1), B, 0.1)
    w1, _ = e.step(x)
    s = e.get_state()
    assert s["has_prev"].tolist() == [1, 1] and s["idx"].tolist() == [1, 1]
    e.reset()
    s0 = e.get_state()
    assert not s0["X"].any() and not s0["U"].any() and not s0["mu"].any() and s0["idx"].tolist() == [0, 0] and s0["has_prev"].tolist() == [0, 0]
    Kx, _ = o.get_kx()
    assert np.abs(s0["C"][0] - Kx).max() < 1e-12           # C_0 = K(X,X) + sn^2 I
    w2, _ = e.step(x)
    assert np.array_equal(w1, w2)                           # same cold start, same answer
    e.set_state(**s)                                        # restore the dump taken after the first step
    s2 = e.get_state()
    for k in s:
        assert np.array_equal(s[k], s2[k]), k


def _work_counters(lib):
    """mpcq_get_qp_work: factorisations | sweeps << 16 of every quadrotor's last solve, consistent with the decimal fields of qp_iter:
    a cold start goes through the interior point (several factorisations, three sweeps per iteration + the start's two + passes), a warm
    success is one factorisation and one forward sweep per pass."""
    from mpc_quad_ros_amd.engine import qp_fallback, qp_passes
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    B = 6
    e = Engine(EngineConfig(batch=B, N=10, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)), lib_path=lib)
    traj, lens = swarm_trajectories(4, 0, B)
    e.set_trajectories(traj, lens); e.sim_reset(np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1)))
    e.sim_steps(1, 2, 5e-3)
    it, (fac, swp) = e.get_qp_iter(), e.get_qp_work()
    assert np.array_equal(fac, qp_passes(it)) and (fac >= 3).all() and (swp >= 3 * (fac - 2)).all()      # cold start: interior point
    e.sim_steps(3, 2, 5e-3)
    it, (fac, swp) = e.get_qp_iter(), e.get_qp_work()
    warm = ~qp_fallback(it)
    assert warm.any() and np.array_equal(fac[warm], qp_passes(it)[warm]) and np.array_equal(swp[warm], fac[warm])
    assert (e.get_qp_float_iterations() == 0).all()      # N = 10 runs the any-shape instance: its interior point iterates in double
    e.close()
    # (20, 10) has a shape-specialised instance whose fallback interior point iterates in float: the work word says how many iterations did
    # (bits 27..31; bench.py prices exactly these with the float chains -- advisor finding of round 5)
    B = 2
    e = Engine(EngineConfig(batch=B, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)), lib_path=lib)
    traj, lens = swarm_trajectories(4, 0, B)
    e.set_trajectories(traj, lens); e.sim_reset(np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1)))
    e.sim_steps(1, 2, 5e-3)
    (fac, swp), fit = e.get_qp_work(), e.get_qp_float_iterations()
    assert (fit >= 2).all() and (fit < fac).all() and (swp >= 3 * fit).all() and not e.get_qp_float_breakdown().any()
    e.sim_steps(2, 2, 5e-3)
    assert (e.get_qp_float_iterations()[~qp_fallback(e.get_qp_iter())] == 0).all()
    e.close()


def _argument_errors(lib):
    with pytest.raises(_lib.MpcqError):
        Engine(EngineConfig(batch=0, N=5), lib_path=lib)
    e = Engine(EngineConfig(batch=1, N=5), lib_path=lib)
    with pytest.raises(_lib.MpcqError, match="set_trajectories"):
        e.step(np.zeros((1, 13)))
    with pytest.raises(_lib.MpcqError):
        e.get_x(6)
    with pytest.raises(_lib.MpcqError, match="RGP"):
        e.rgp_regress(np.zeros((1, 3)), np.zeros((1, 3)))
    with pytest.raises(ValueError):
        e.set_trajectories(np.zeros((2, 10, 13)))
    with pytest.raises(_lib.MpcqError):
        e.set_trajectories(np.zeros((1, 10, 13)), np.array([11]))
    # solver tuning is validated, not read from the environment
    for bad in (dict(warm_max=-3), dict(warm_max=1000), dict(pin_ratio=-1.0), dict(ipm_mu0=5.0), dict(ipm_margin=0.7), dict(stage_mem=4),
                dict(flip_max=-2), dict(ipm_tol=float("nan")), dict(block_order=3), dict(groups=17), dict(groups=-1)):
        with pytest.raises(_lib.MpcqError, match="tune"):
            Engine(EngineConfig(batch=1, N=5, tune=bad), lib_path=lib)
    Engine(EngineConfig(batch=1, N=5, tune=dict(warm_max=8, flip_max=-1, abort_pins=-1, stage_mem="global", pin_ratio=0.5)), lib_path=lib).close()
    Engine(EngineConfig(batch=1, N=5, tune=dict(stage_mem="compact", block_order=2)), lib_path=lib).close()
    Engine(EngineConfig(batch=1, N=5, tune=dict(groups=4)), lib_path=lib).close()      # (a batch too small for four groups runs as one)
    # advisor finding of round 5: without the active-set passes a float engine would answer from the float interior point (status 8 on every fallback solve)
    with pytest.raises(_lib.MpcqError, match="PRECISION_F32"):
        Engine(EngineConfig(batch=1, N=5, precision=1, tune=dict(polish_max=-1)), lib_path=lib)
    Engine(EngineConfig(batch=1, N=5, precision=0, tune=dict(polish_max=-1)), lib_path=lib).close()
    # a restored checkpoint is validated (advisor finding of round 3: a negative cursor used to reach the kernel)
    traj = np.zeros((1, 10, 13)); traj[:, :, 3] = 1.0
    e.set_trajectories(traj)
    for bad_state, what in ((dict(idx=np.array([-7])), "cursor"), (dict(has_prev=np.array([2])), "has_prev")):
        with pytest.raises(_lib.MpcqError, match=what):
            e.set_state(**bad_state)
    e.set_state(idx=np.array([14]), has_prev=np.array([1]))      # at or beyond the end is a valid cursor (the chunk repeats the last row)
    for bad_sol, what in ((dict(qp_iter=np.array([-1])), "qp_iter"), (dict(qp_iter=np.array([1000000])), "qp_iter"), (dict(finished=np.array([3])), "finished")):
        with pytest.raises(_lib.MpcqError, match=what):
            e.set_solver_state(**bad_sol)
    with pytest.raises(ValueError, match="unknown tuning field"):
        EngineConfig(batch=1, N=5, tune=dict(warm=3)).to_c()
    _versioned_create(lib)


def _versioned_create(lib_path):
    """mpcq_create_sized: a caller built against an older header passes ITS struct size; fields behind it take their defaults,
    garbage behind the declared size is never read; sizes the library does not understand are refused."""
    import ctypes
    from mpc_quad_ros_amd.params import CConfig
    lib = _lib.load(lib_path)
    c = EngineConfig(batch=2, N=5).to_c()
    raw = (ctypes.c_char * (ctypes.sizeof(CConfig) + 64))()
    ctypes.memmove(raw, ctypes.byref(c), ctypes.sizeof(CConfig))
    old_size = CConfig.finish_radius.offset                      # the 0.1 layout: ends before finish_radius
    ctypes.memset(ctypes.addressof(raw) + old_size, 0xAB, ctypes.sizeof(CConfig) + 64 - old_size)   # garbage where newer fields would be
    h = ctypes.c_void_p()
    assert lib.mpcq_create_sized(ctypes.cast(raw, ctypes.c_void_p), old_size, ctypes.byref(h)) == 0, lib.mpcq_last_error()
    lib.mpcq_destroy(h)
    v03 = CConfig.tune.offset                                    # the 0.2 layout: ends behind finish_radius
    c2 = EngineConfig(batch=2, N=5, finish_radius=0.5).to_c()
    ctypes.memmove(raw, ctypes.byref(c2), v03)
    assert lib.mpcq_create_sized(ctypes.cast(raw, ctypes.c_void_p), v03, ctypes.byref(h)) == 0, lib.mpcq_last_error()
    lib.mpcq_destroy(h)
    for size in (CConfig.device.offset, ctypes.sizeof(CConfig) + 8):
        assert lib.mpcq_create_sized(ctypes.cast(raw, ctypes.c_void_p), size, ctypes.byref(h)) != 0
        assert b"size" in lib.mpcq_last_error()
    assert b"0.6" in lib.mpcq_version()


def _reference_format_log(lib):
    from mpc_quad_ros_amd.logger import REFERENCE_KEYS, SwarmLogger
    from helpers import load_golden
    B, N, nb = 2, 5, 10
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb)), lib_path=lib)
    traj = np.zeros((B, 40, 13)); traj[:, :, 3] = 1.0; traj[:, :, 2] = 3.0; traj[:, :, 0] = np.linspace(0, 2, 40)
    e.set_trajectories(traj)
    lg = SwarmLogger(e)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    for k in range(4):
        w, xp = e.step(x)
        lg.log_step(0.01 * k, x, w, xp, traj[:, k])
        x = xp
    q = lg.quad_log(1)
    g = load_golden("log_traj0_v10_a10_gp2.npz")
    assert tuple(q.keys()) == REFERENCE_KEYS
    assert q["x_odom"].shape == (4, 13) and q["w_odom"].shape == (4, 4) and q["t_cpu"].shape == (4, 1)
    assert q["rgp_mu_g_t"].shape == (4, 3, nb) and q["rgp_C_g_t"].shape == (4, 3, nb, nb) and q["v_body"].shape == (4, 3, 1)
    assert q["rgp_mu_g_t"].shape[1:] == g["rgp_mu"].shape[1:]        # same per-step layout as the reference's pickles
    st = e.get_tracking_stats()
    rms = lg.rms_position_error()
    assert np.isclose(np.sqrt(st[0] / (3 * st[2])), np.sqrt(np.mean(rms ** 2)), rtol=1e-9)
    # the CPU-time summary of Visualiser.plot_data (src/Visualiser.py:981-987) over the same log
    avg, std, per_quad = lg.cpu_time_summary()
    t_cpu = np.array([q["t_cpu"][k, 0] for k in range(4)])
    assert np.isclose(avg, np.mean(t_cpu)) and np.isclose(std, np.std(t_cpu)) and np.isclose(per_quad, avg / B) and avg >= 0
    assert lg.summary_title().startswith("MPC CPU Time, Avg: ")


def _reference_analysis_round_trip(lib):
    """f2 round trip: a run of the engine on the REFERENCE's own measurements (a logged python-sim flight: x_odom fed step by step,
    x_ref = row 0 of every chunk, as the node logs it, src/execute_trajectory.py:270-273) is written by SwarmLogger.save() as a pickle
    in the reference's layout and read back by a restatement of Visualiser.plot_data's numeric path (tests/helpers.visualiser_summaries:
    src/Visualiser.py:787-811,918,981-987).  The total position RMS it prints equals (i) the statistic the device accumulates
    (mpcq_get_tracking_stats, a10) and (ii) the same analysis run on the reference's golden log itself; the CPU-time panel equals the
    logger's own summary."""
    import tempfile
    from mpc_quad_ros_amd.logger import SwarmLogger
    from helpers import config_for_log, load_golden, visualiser_summaries
    g = load_golden("log_traj0_v10_a10_gp2.npz")
    K = 12 if lib is not None else 100
    e = Engine(config_for_log(g), lib_path=lib)
    e.set_trajectories(g["x_ref"][None])
    lg = SwarmLogger(e)
    for k in range(K):
        x = g["x_odom"][k][None]
        chunk0 = e.get_reference_chunk()[:, 0]            # row 0 of this step's chunk (get_reference_chunk at the current cursor)
        assert np.array_equal(chunk0[0], g["x_ref"][k])   # = what the reference logged as x_ref
        w, xp = e.step(x)
        lg.log_step(0.1 * k, x, w, xp, chunk0)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "run.pkl")
        lg.save(path, 0)
        mine = visualiser_summaries(path)
    ref = visualiser_summaries({"x_odom": g["x_odom"][:K], "x_ref": g["x_ref"][:K], "t_cpu": np.zeros((K, 1))})
    st = e.get_tracking_stats()
    assert st[2] == K
    assert np.isclose(mine["rms_total"], np.sqrt(st[0] / (3 * st[2])), rtol=1e-12, atol=0)          # the device's accumulators
    assert mine["rms_total"] == ref["rms_total"] and np.array_equal(mine["rms_pos_ref"], ref["rms_pos_ref"])   # the reference's own log
    assert np.isclose(np.sqrt(st[3]), np.sqrt(3.0) * mine["rms_pos_ref"].max(), rtol=1e-12)          # max |e_pos| = sqrt(3) x the largest per-step RMS
    avg, std, _ = lg.cpu_time_summary()
    assert mine["avg_cpu"] == avg and mine["std_cpu"] == std and mine["title_cpu"] == lg.summary_title()
    assert mine["title_rms"].startswith("RMS Position Error, Total: ") and mine["title_rms"].endswith("mm")


def _free_running_equals_lockstep(lib):
    """mpcq_sim_run (one launch, every instance runs its K periods on its own) must reproduce mpcq_sim_steps
    (one launch per period) bit for bit: state, controls, cursors, RGP posterior, tracking statistic.  Ragged
    trajectory lengths so that some instances run past the end of their reference inside the launch."""
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    big = lib is None
    B, N, nb, K = (64, 20, 10, 40) if big else (3, 20, 10, 9)
    traj, lens = swarm_trajectories(5, 0, B)
    lens = lens.copy(); lens[0] = 6
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    for precision in (0, 1):
        res = []
        for mode in ("sim_steps", "sim_run"):
            e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=precision), lib_path=lib)
            e.set_trajectories(traj, lens); e.sim_reset(x0)
            getattr(e, mode)(K, 2, 5e-3)
            getattr(e, mode)(2, 2, 5e-3)            # a second call continues where the first stopped
            st = e.get_state()
            mu, C = e.get_rgp()
            res.append([*e.sim_get_state(), st["X"], st["U"], st["idx"], mu, C, e.get_tracking_stats(), e.get_status(), e.get_cost()])
            e.close()
        for a, b in zip(*res):      # bit for bit in both precisions (both launch modes run the shape-specialised instance of the precision)
            assert np.array_equal(np.asarray(a), np.asarray(b))
        assert (res[0][4] == K + 2).all()


def _command_and_finished(lib):
    """a4: the command mapping of publish_control_gazebo (src/mpc_controller_node.py:600-612), bit-exact against the
    numpy expressions; a9: the finished predicate (src/mpc_controller_node.py:374) against the oracle's flags."""
    B, N, nb = 4, 5, 10
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), skip=1)
    e, o = engines(lib, **kw)
    T = 12
    traj = np.zeros((B, T, 13)); traj[:, :, 3] = 1.0; traj[:, :, 2] = 3.0
    traj[:, :, 0] = np.linspace(0, 0.3, T)[None, :]
    traj[3, :, 0] += 5.0                                  # quadrotor 3 never gets within 1 m of its reference: never finishes
    lens = np.array([T, T - 3, 6, T], dtype=np.int32)     # different lengths: flags rise at different steps
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    q = hummingbird()
    seen = []
    for k in range(T + 2):
        w, xp = e.step(x); wo, xpo = o.step(x)
        rotor, coll, rates = e.get_command()
        assert np.array_equal(rotor, w * q.max_thrust / q.mass)
        assert np.array_equal(coll, np.sum(w, axis=1) * q.max_thrust / q.mass)
        assert np.array_equal(rates, e.get_x(1)[:, 10:13])
        ro, co, rao = o.get_command()
        assert np.abs(rotor - ro).max() < 1e-6 and np.abs(coll - co).max() < 1e-6 and np.abs(rates - rao).max() < 1e-6
        assert np.array_equal(e.get_finished(), o.get_finished()), k
        seen.append(e.get_finished().copy())
        x = xpo
    seen = np.array(seen)
    # idx_traj + 1 == len is tested after the increment: the flag rises in the step whose cursor was len - 2, and stays
    for b in range(3):
        assert seen[:, b].tolist() == [0] * (lens[b] - 2) + [1] * (T + 2 - (lens[b] - 2))
    assert not seen[:, 3].any()
    e.set_trajectories(traj, lens)
    assert not e.get_finished().any()                     # a new trajectory clears the flag (trajectory_received_cb)


def _chunk_cases_on_device(lib):
    """All 1 239 get_reference_chunk cases generated by executing the reference's function
    (tests/golden/utils_vectors.npz), pushed through the device-side row selection the fused step uses."""
    from helpers import load_golden
    v = load_golden("utils_vectors.npz")
    rng = np.random.default_rng(0)
    groups = {}
    for (T, N, skip, idx), rows in zip(v["chunk_cases"], v["chunk_rows"]):
        groups.setdefault((int(N), int(skip)), []).append((int(T), int(idx), rows))
    total = 0
    for (N, skip), cases in sorted(groups.items()):
        B, Tmax = len(cases), max(c[0] for c in cases)
        e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), skip=skip), lib_path=lib)
        traj = rng.normal(size=(B, Tmax, 13))
        lens = np.array([c[0] for c in cases], dtype=np.int32)
        e.set_trajectories(traj, lens)
        e.set_state(idx=np.array([c[1] for c in cases], dtype=np.int32))
        ch = e.get_reference_chunk()
        for b, (T, idx, rows) in enumerate(cases):
            assert np.array_equal(ch[b], traj[b][rows[:N]]), (T, N, skip, idx)
        total += B
        e.close()
    assert total == len(v["chunk_cases"]) == 1239


def _plant_period_matches_reference_logs(lib):
    """f1: the device plant with the reference's float-accumulated substep loop (src/execute_trajectory.py:232-243)
    against the logged python-simulation states: x_odom[k+1] = plant(x_odom[k], w_odom[k]), 20 substeps of 5 ms."""
    from helpers import config_for_log, load_golden
    for name, K in (("log_traj1_v10_a10_gp0.npz", 120), ("log_traj0_v10_a10_gp2.npz", 100)):
        g = load_golden(name)
        cfg = config_for_log(g, batch=K)
        e = Engine(cfg, lib_path=lib)
        assert [e.plant_substeps(d) for d in (0.1, 0.05, 0.02)] == [20, 11, 4]
        e.sim_reset(g["x_odom"][:K])
        n = e.sim_plant_period(g["w_odom"][:K], cfg.optimization_dt, 5e-3)
        assert n == 20
        x, w = e.sim_get_state()
        assert np.abs(x - g["x_odom"][1:K + 1]).max() < 1e-12
        assert np.array_equal(w, g["w_odom"][:K])
        o = OracleEngine(config_for_log(g, batch=K))
        xo, no = o.plant_control_period(g["x_odom"][:K], g["w_odom"][:K], cfg.optimization_dt, 5e-3)
        assert no == n and np.abs(x - xo).max() < 1e-13
        e.close()


def _checkpoint_resume_is_bitwise(lib):
    """get_state + get_solver_state + sim_get_state is the whole resumable state: an engine restored from a dump
    continues bit for bit (warm-start flag, tracking accumulators and finished flags included)."""
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    B, N, nb, K1, K2 = (32, 20, 10, 30, 25) if lib is None else (2, 10, 10, 4, 3)
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb))
    traj, lens = swarm_trajectories(9, 0, B)
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    a = Engine(EngineConfig(**kw), lib_path=lib)
    a.set_trajectories(traj, lens); a.sim_reset(x0)
    a.sim_control_periods(K1, 0.01, 5e-3)
    dump = (a.get_state(), a.get_solver_state(), a.sim_get_state()[0])
    assert (dump[1]["qp_iter"] > 0).all() and (dump[1]["stats"][:, 2] == K1).all()
    a.sim_control_periods(K2, 0.01, 5e-3)
    b = Engine(EngineConfig(**kw), lib_path=lib)
    b.set_trajectories(traj, lens)
    b.set_state(**dump[0]); b.set_solver_state(**dump[1]); b.sim_reset(dump[2])
    b.sim_control_periods(K2, 0.01, 5e-3)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    for k, va in a.get_solver_state().items():
        assert np.array_equal(va, b.get_solver_state()[k]), k
    assert np.array_equal(a.sim_get_state()[0], b.sim_get_state()[0]) and np.array_equal(a.sim_get_state()[1], b.sim_get_state()[1])
    assert np.array_equal(a.get_tracking_stats(), b.get_tracking_stats())


def _rgp_learn_matches_reference_streams(lib):
    """f4: batched RGP.learn on the device (mpcq_learn_*) against the vectors produced by importing the reference's
    RGP.py (tests/golden/learn_vectors.npz) and against the CPU restatement on a batch of different streams."""
    from helpers import load_golden
    from mpc_quad_ros_amd.engine import Learner
    from oracle.oracle import OracleLearner
    v = load_golden("learn_vectors.npz")
    for c in range(int(v["ncases"])):
        p = f"c{c}_"
        nb, X, theta = int(v[p + "nb"]), v[p + "X"], v[p + "theta"]
        lr = Learner(2, np.tile(X, (3, 1)), theta, lib_path=lib)
        s, y = v[p + "s"], v[p + "y"]
        for k in range(len(s)):
            lr.step(np.full((2, 3), s[k]), np.full((2, 3), y[k]))
            g = lr.get()
            mu_z = np.concatenate([g["mu_g"][1, 2], g["mu_eta"][1, 2]])
            assert np.abs(mu_z - v[p + "mu_z"][k]).max() < 1e-8 * max(1.0, np.abs(v[p + "mu_z"][k]).max()), (c, k)
        Cz, Ki = v[p + "C_z"][-1], v[p + "K_x_inv_last"]
        assert np.abs(g["C_g"][0, 0] - Cz[:nb, :nb]).max() < 1e-8 * max(1.0, np.abs(Cz).max())
        assert np.abs(g["C_eta"][0, 0] - Cz[nb:, nb:]).max() < 1e-8 * max(1.0, np.abs(Cz).max())
        assert np.abs(g["K_x_inv"][1, 1] - Ki).max() < 1e-7 * np.abs(Ki).max()
        lr.close()
    # a batch of different streams per regressor against the CPU restatement
    rng = np.random.default_rng(3)
    B, nb = 5, 12
    basis = np.tile(np.linspace(-12, 12, nb), (3, 1))
    theta = np.array([[1.0, 0.1, 0.1], [2.0, 0.5, 0.05], [1.5, 0.3, 0.2]])
    a, o = Learner(B, basis, theta, lib_path=lib), OracleLearner(B, basis, theta)
    for k in range(12):
        sv = rng.uniform(-10, 10, (B, 3)); yv = 0.3 * sv + rng.normal(0, 0.1, (B, 3))
        a.step(sv, yv); o.step(sv, yv)
    ga, go = a.get(), o.get()
    for key in ga:
        assert np.abs(ga[key] - go[key]).max() < 1e-8 * max(1.0, np.abs(go[key]).max()), key
    with pytest.raises(_lib.MpcqError):
        Learner(1, np.zeros((3, 65)), [1.0, 0.1, 0.1], lib_path=lib)


def _static_gp_model_path(lib):
    """use_gp = 1: static GP in the model (flag MPCQ_FLAG_STATIC_GP): training inputs as basis, responses loaded with
    set_params, no recursive update in the fused step; against the oracle in the same mode."""
    from helpers import load_golden
    from mpc_quad_ros_amd.params import static_gp_theta
    v = load_golden("gp_vectors.npz")
    X, y, theta = v["c0_X"], v["c0_y"], v["c0_theta"]
    n, B, N = len(X), 3, 10
    kw = dict(batch=B, N=N, quad=hummingbird(), nb=n, basis=np.tile(X, (3, 1)), theta=static_gp_theta(theta))
    e = Engine(EngineConfig(static_gp=True, **kw), lib_path=lib)
    o = OracleEngine(EngineConfig(**kw)); o.set_static_gp(True)
    mu = np.tile(np.tile(y, 3), (B, 1))
    traj = np.zeros((B, 60, 13)); traj[:, :, 3] = 1.0; traj[:, :, 2] = 3.0
    traj[:, :, 0] = np.linspace(0, 4, 60)[None, :] * np.array([1.0, 1.5, 2.0])[:, None]
    traj[:, :, 7] = (4 / 0.59) * np.array([1.0, 1.5, 2.0])[:, None]
    for eng in (e, o):
        eng.set_trajectories(traj); eng.set_params(mu)
    x = traj[:, 0].copy()
    for k in range(8):
        w, xp = e.step(x); wo, xpo = o.step(x)
        assert np.abs(w - wo).max() < 1e-8 and (e.get_status() == 0).all()
        x = o.plant_control_period(x, wo, 0.01, 5e-3)[0]
    mu_e, C_e = e.get_rgp(); mu_o, C_o = o.get_rgp()
    assert np.array_equal(mu_e.reshape(B, -1), mu) and np.array_equal(mu_o.reshape(B, -1), mu)      # untouched by the loop
    Kx, _ = o.get_kx()
    assert np.abs(C_e[0] - Kx).max() < 1e-12
    # the same engine without the flag does update
    e2 = Engine(EngineConfig(**kw), lib_path=lib)
    e2.set_trajectories(traj); e2.set_params(mu); e2.step(traj[:, 0].copy())
    assert not np.array_equal(e2.get_rgp()[0].reshape(B, -1), mu)


CASES = [_ragged_and_exhausted, _reset_and_state_roundtrip, _argument_errors, _reference_format_log, _reference_analysis_round_trip, _free_running_equals_lockstep,
         _command_and_finished, _chunk_cases_on_device, _plant_period_matches_reference_logs, _checkpoint_resume_is_bitwise,
         _rgp_learn_matches_reference_streams, _static_gp_model_path, _work_counters]


@pytest.mark.parametrize("case", CASES, ids=[c.__name__.strip("_") for c in CASES])
def test_emu(case):
    case(EMU)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c.__name__.strip("_") for c in CASES])
def test_gpu(case):
    case(None)
