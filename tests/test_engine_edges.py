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
        for a, b in zip(*res):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        assert (res[0][4] == K + 2).all()


CASES = [_ragged_and_exhausted, _reset_and_state_roundtrip, _argument_errors, _reference_format_log, _free_running_equals_lockstep]


@pytest.mark.parametrize("case", CASES, ids=[c.__name__.strip("_") for c in CASES])
def test_emu(case):
    case(EMU)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c.__name__.strip("_") for c in CASES])
def test_gpu(case):
    case(None)
