"""Multi-rank path on CPU (gloo, world_size 2): the swarm is partitioned by global index, per-rank
results are partition-invariant, and the tracking statistic reduces like the RCCL path does
(SUM of [sum e_pos^2, sum e_vel^2, steps, #failed], MAX of max e_pos^2).  The compute engine here is
the CPU oracle (tests may use it); the product path uses the same partition helpers + RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, K, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
    from mpc_quad_ros_amd.trajectories import shard_range, swarm_trajectories
    from oracle.oracle import OracleEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, n = shard_range(total, rank, world)
    e = OracleEngine(EngineConfig(batch=n, N=10, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)))
    traj, lens = swarm_trajectories(11, lo, n)
    e.set_trajectories(traj, lens)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (n, 1))
    ws = []
    for _ in range(K):
        w, xp = e.step(x)
        ws.append(w)
        x = xp
    s = e.get_tracking_stats()
    local = np.array([s[0], s[1], s[2], s[3], 0.0])
    tsum = torch.tensor([local[0], local[1], local[2], 0.0, local[4]], dtype=torch.float64)
    tmax = torch.tensor([local[3]], dtype=torch.float64)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    red = tsum.numpy().copy()
    red[3] = tmax.item()
    q.put((rank, lo, n, np.stack(ws), local, red))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_index_once():
    from mpc_quad_ros_amd.trajectories import shard_range
    for total in (1, 7, 8, 1024, 65536, 1000):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
            for (lo, n), (lo2, _) in zip(blocks, blocks[1:]):
                assert lo + n == lo2
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1


def test_trajectories_depend_only_on_seed_and_global_index():
    from mpc_quad_ros_amd.trajectories import swarm_trajectories
    a, la = swarm_trajectories(5, 0, 6)
    b, lb = swarm_trajectories(5, 3, 3)
    assert np.array_equal(la[3:], lb)
    for i in range(3):
        assert np.array_equal(a[3 + i, :lb[i]], b[i, :lb[i]])
    c, _ = swarm_trajectories(6, 0, 1)
    assert not np.array_equal(a[0, :50], c[0, :50])


def test_world_size_2_gloo_matches_single_rank():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    total, K, port = 6, 4, _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, K, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-rank reference of the same swarm
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_worker, args=(0, 1, _free_port(), total, K, q1))
    p1.start()
    _, _, _, w_all, _, red1 = q1.get(timeout=120)
    p1.join(timeout=60)
    w_sharded = np.concatenate([r[3] for r in res], axis=1)
    assert np.array_equal(w_sharded, w_all)                    # partition-invariant results
    for r in res:
        assert np.allclose(r[5], red1, rtol=1e-12, atol=0)     # every rank holds the swarm statistic
    assert res[0][5][2] == total * K


def test_circle_references_match_reference_generator():
    """Closed-form circle trajectories (cheap half) against vectors produced by the reference's
    TrajectoryGenerator (tests/golden/make_golden.py: make_circle_vectors)."""
    from mpc_quad_ros_amd.trajectories import circle_trajectory
    g = np.load(os.path.join(ROOT, "tests", "golden", "circle_vectors.npz"))
    x, t = circle_trajectory("accelerating", 10, 12, dt=0.1, t_max=30, start_point=(0.0, 0.0, 3.0))
    assert x.shape == g["acc_x"].shape and np.abs(x - g["acc_x"]).max() <= 1.5e-6 and np.abs(t - g["acc_t"]).max() < 1e-9
    x, t = circle_trajectory("constant", 5.0, 8.0, dt=0.05, start_point=(1.0, -2.0, 3.0))
    assert x.shape == g["const_x"].shape and np.abs(x - g["const_x"]).max() <= 1.5e-6
    x, t = circle_trajectory("acc_dec", 10, 10, dt=0.01)
    assert x.shape == g["ad_x"].shape and np.abs(x - g["ad_x"]).max() <= 1.5e-6


def test_polynomial_references_match_reference_sampler():
    """Piecewise-polynomial references (the output format of the reference's min-snap generator) sampled by
    trajectories.sample_polynomial_trajectory against vectors from the reference's own uav_trajectory evaluator +
    TrajectoryGenerator.save_evals_csv / load_trajectory (tests/golden/make_golden.py: make_poly_vectors)."""
    from mpc_quad_ros_amd.trajectories import sample_polynomial_trajectory
    g = np.load(os.path.join(ROOT, "tests", "golden", "poly_vectors.npz"))
    for c in range(int(g["ncases"])):
        x, t = sample_polynomial_trajectory(g[f"p{c}_pieces"], float(g[f"p{c}_dt"]))
        assert x.shape == g[f"p{c}_x"].shape
        assert np.abs(x - g[f"p{c}_x"]).max() <= 1.5e-6 and np.abs(t - g[f"p{c}_t"]).max() < 1e-9
