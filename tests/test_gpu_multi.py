"""Everything about the multi-GPU path that ONE MI355X allows (BASELINE configs[3]):
the RCCL reduction of the swarm statistic through mpcq_comm_* on the real librccl.so (1-rank communicator), partition
invariance of the sharding on the device, configs[3]'s per-rank shape against the oracle, and the device-buffer step.
Run with `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import shard_range, swarm_trajectories

pytestmark = pytest.mark.gpu
X0 = np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def swarm_engine(B, first=0, seed=2026, N=20, nb=10, precision=0):
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=precision))
    traj, lens = swarm_trajectories(seed, first, B)
    e.set_trajectories(traj, lens)
    e.sim_reset(np.tile(X0, (B, 1)))
    return e, traj, lens


def test_rccl_allreduce_single_rank_communicator():
    """mpcq_comm_unique_id -> mpcq_comm_init(rank 0 of 1) -> mpcq_allreduce_tracking_stats on the real RCCL:
    the reduced vector equals the local statistic (SUM of slots 0,1,2,4, MAX of slot 3)."""
    e, _, _ = swarm_engine(256)
    e.sim_steps(30, 2, 5e-3)
    local = e.get_tracking_stats()
    uid = e.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e.comm_init(0, 1, uid)
    red = e.allreduce_tracking_stats()
    assert np.array_equal(red, local) and red[2] == 256 * 30 and red[3] > 0
    e.sim_steps(5, 2, 5e-3)                        # the communicator stays usable across further steps
    assert np.array_equal(e.allreduce_tracking_stats(), e.get_tracking_stats())
    with pytest.raises(Exception, match="already"):
        e.comm_init(0, 1, uid)


def test_partition_invariance_on_the_device():
    """Two engines holding the shards [0,512) and [512,1024) of a swarm reproduce one 1 024-instance engine bit for
    bit (controls, plant states, iterate, RGP posterior), and their statistics combine like the RCCL reduction."""
    total, K = 1024, 60
    full, _, _ = swarm_engine(total)
    full.sim_steps(K, 2, 5e-3)
    xf, wf = full.sim_get_state()
    sf = full.get_state()
    stf = full.get_tracking_stats()
    parts = []
    for rank in range(2):
        lo, n = shard_range(total, rank, 2)
        e, _, _ = swarm_engine(n, first=lo)
        e.sim_steps(K, 2, 5e-3)
        parts.append((lo, n, e.sim_get_state(), e.get_state(), e.get_tracking_stats()))
    for lo, n, (x, w), s, _ in parts:
        assert np.array_equal(x, xf[lo:lo + n]) and np.array_equal(w, wf[lo:lo + n])
        for k in ("X", "U", "mu", "C", "idx"):
            assert np.array_equal(s[k], sf[k][lo:lo + n]), k
    a, b = parts[0][4], parts[1][4]
    assert a[2] + b[2] == stf[2] and max(a[3], b[3]) == stf[3] and a[4] + b[4] == stf[4]
    assert np.isclose(a[0] + b[0], stf[0], rtol=1e-13) and np.isclose(a[1] + b[1], stf[1], rtol=1e-13)


def test_config3_per_rank_shape_against_oracle():
    """BASELINE configs[3]: 65 536 quadrotors over 8 GPUs = 8 192 per rank, N=20, nb=10.  Rank 5's shard
    (global indices 40 960 ...) for 40 closed-loop periods on the device; a random sample of 64 of its instances
    against the oracle driven through the same closed loop (own plant) on the host."""
    from oracle.oracle import OracleEngine
    total, world, rank, K = 65536, 8, 5, 40
    lo, n = shard_range(total, rank, world)
    assert n == 8192
    e, traj, lens = swarm_engine(n, first=lo)
    e.sim_control_periods(K, 0.01, 5e-3)
    assert (e.get_status() == 0).all()
    xe, we = e.sim_get_state()
    st = e.get_tracking_stats()
    assert st[2] == n * K and st[4] == 0
    pick = np.sort(np.random.default_rng(3).choice(n, 64, replace=False))
    o = OracleEngine(EngineConfig(batch=64, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)))
    o.set_trajectories(traj[pick], lens[pick])
    x = np.tile(X0, (64, 1))
    for _ in range(K):
        w, _ = o.step(x)
        x, nsub = o.plant_control_period(x, w, 0.01, 5e-3)
    assert nsub == 2
    assert np.abs(we[pick] - w).max() < 1e-6 and np.abs(xe[pick] - x).max() < 1e-6
    assert np.array_equal(e.get_state()["idx"], np.full(n, K))


def test_block_order_changes_nothing():
    """Large batches run their lockstep periods in the launch order of mpcq::order_kernel (quadrotors predicted expensive
    first, mpcq_tuning.block_order).  B = 2 560 (more than the device holds at once, not a multiple of the class structure's
    natural sizes) on the bench workload, in flight: the ordered launches equal the identity-ordered ones bit for bit, the
    order is a permutation inside the classes p mod 8, each class in ascending cost bin of the previous period.  Since round 6 such a batch
    also runs as two groups on streams of their own (mpcq_tuning.groups = 0: automatic), each sorted on its own: the reference engine here
    is the plain one -- one group, identity order."""
    import bench
    from mpc_quad_ros_amd.engine import order_bin
    B, pre, K = 2560, 150, 25
    refs = bench.workload(2026, 0, B, pre + K + 1)
    engines = []
    for bo in (1, 0):      # 0 = automatic: on, since B exceeds the resident capacity
        e = Engine(EngineConfig(batch=B, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10), tune=dict(block_order=bo, groups=bo)))
        e.set_trajectories(*refs); e.sim_reset(np.tile(X0, (B, 1)))
        e.sim_run(pre, 2, 5e-3)
        e.sim_steps(K, 2, 5e-3)
        engines.append(e)
    a, b = engines
    G = b.get_groups()
    assert a.get_groups() == 1 and G == 2
    per = ((B + G - 1) // G + 7) // 8 * 8             # group g = quadrotors [g per, (g + 1) per) (mpcq_api.hip: sim_steps)
    it_prev = b.get_qp_iter()
    assert len(np.unique(order_bin(it_prev))) >= 3          # the workload spans several cost bins here
    for e in engines:
        e.sim_steps(1, 2, 5e-3)
    assert np.array_equal(a.get_block_order(), np.arange(B))
    order = b.get_block_order()
    assert np.array_equal(np.sort(order), np.arange(B))
    for g0 in range(0, B, per):
        seg = order[g0:min(B, g0 + per)]
        assert np.array_equal(np.sort(seg), np.arange(g0, min(B, g0 + per)))      # a permutation of the group's own quadrotors
        for x in range(8):
            cls = seg[x::8]
            assert np.all(cls % 8 == x)
            bins = order_bin(it_prev[cls])
            assert np.all(np.diff(bins) >= 0)
    (xa, wa), (xb, wb) = a.sim_get_state(), b.sim_get_state()
    assert np.array_equal(xa, xb) and np.array_equal(wa, wb)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.get_qp_iter(), b.get_qp_iter()) and np.array_equal(a.get_tracking_stats(), b.get_tracking_stats())


class _Hip:
    """Raw device buffers for the device-pointer entry point (no torch on this path)."""
    def __init__(self):
        self.lib = ctypes.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.lib.hipFree.argtypes = [ctypes.c_void_p]

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        assert self.lib.hipMalloc(ctypes.byref(p), nbytes) == 0
        return p.value

    def h2d(self, dst, a):
        a = np.ascontiguousarray(a)
        assert self.lib.hipMemcpy(ctypes.c_void_p(dst), a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 1) == 0

    def d2h(self, a, src):
        assert self.lib.hipMemcpy(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src), a.nbytes, 2) == 0

    def free(self, p):
        self.lib.hipFree(ctypes.c_void_p(p))


@pytest.mark.parametrize("precision", [0, 1])
def test_step_device_async_matches_host_step(precision):
    """mpcq_step_device_async takes float64 device buffers in both precisions and equals mpcq_step bit for bit."""
    hip = _Hip()
    B = 128
    a, traj, lens = swarm_engine(B, precision=precision)
    b, _, _ = swarm_engine(B, precision=precision)
    dx, dw = hip.alloc(B * 13 * 8), hip.alloc(B * 4 * 8 + 64)
    guard = np.full(8, 777.0)
    hip.h2d(dw + B * 4 * 8, guard)
    x = np.tile(X0, (B, 1))
    x[:, :3] += np.random.default_rng(2).normal(0, 0.05, (B, 3))
    for k in range(6):
        w_host, xp = a.step(x)
        hip.h2d(dx, x)
        b.step_device_async(dx, dw)
        b.synchronize()
        w_dev = np.zeros((B, 4))
        hip.d2h(w_dev, dw)
        assert np.array_equal(w_dev, w_host), k
        x = xp
    tail = np.zeros(8)
    hip.d2h(tail, dw + B * 4 * 8)
    assert np.array_equal(tail, guard)              # nothing written past [B,4] doubles
    for k, v in a.get_state().items():
        assert np.array_equal(v, b.get_state()[k]), k
    # the engine's own control record follows a step with a caller-supplied control buffer: command mapping and plant agree
    for ca, cb in zip(a.get_command(), b.get_command()):
        assert np.array_equal(ca, cb)
    hip.free(dx); hip.free(dw)


def test_get_x_get_u_all_stages_equal_state_dump():
    e, _, _ = swarm_engine(64)
    e.sim_steps(3, 2, 5e-3)
    s = e.get_state()
    for i in range(21):
        assert np.array_equal(e.get_x(i), s["X"][:, i])
    for i in range(20):
        assert np.array_equal(e.get_u(i), s["U"][:, i])


def test_facade_mirrors_quad_optimizer_on_the_gpu():
    import test_emu_parity as t
    t.test_emu_facade_mirrors_quad_optimizer(lib=None)
