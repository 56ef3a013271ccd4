"""bench.py contract on CPU: argument surface, one JSON line with the required keys, and the N>1 launch
through torch.distributed.run (world_size 2, gloo) using the lane emulator in place of the GPU library."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQ = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
       "vs_baseline", "dtype", "data", "config", "roofline"}


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "wave_emu")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "fake_rccl")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _check(line, n):
    d = json.loads(line)
    assert REQ <= set(d), REQ - set(d)
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] in ("f64", "f32") and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 0 and d["tracking"]["steps"] == d["config"]["global_batch"] * (d["steps"] + d["warmup"] + d["config"]["preroll_periods"])
    return d


def test_single_rank_line():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "run_bench_emu.py"), "--steps", "2", "--warmup", "1", "--preroll", "3",
                                   "--batch", "2", "--horizon", "5", "--nb", "10", "--no-cpu-baseline", "--no-alt", "--steady", "3"], cwd=ROOT)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 1)
    # the oracle-parity leg of the line (here: emulated lanes vs the oracle on the tiny workload)
    p = d["parity_on_workload"]
    assert p["quad_steps"] == 2 * 30 and p["max_rel_dev"] < 1e-7 and p["failed"] == 0
    assert d["roofline"]["traffic"] is None and "no PMC profile" in d["roofline"]["traffic_note"]
    # the representative rate next to the driver's window, and the per-rank record (one rank: min = mean = max = the line's own)
    assert d["steady_state"]["steps"] == 3 and d["steady_state"]["value"] > 0
    pr = d["per_rank"]
    assert pr["steps_per_s"]["min"] == pr["steps_per_s"]["max"] and len(pr["ranks"]) == 1 and 0.5 < d["efficiency_vs_best_rank"] <= 1.0 + 1e-9
    # the latency floor of a launch: chain latencies from profiles/r5_chain_floor.json x the work counters read back after every launch
    lr = d["latency_roofline"]
    assert lr["launches"] == 20 and lr["floor_ms"] > 0 and abs(lr["frac"] - lr["floor_ms"] / lr["achieved_ms"]) < 1e-9 and lr["slowest_quad_factorisations_mean"] >= 1


def test_two_ranks_under_torchrun():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_bench_emu.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--preroll", "3",
           "--batch", "2", "--horizon", "5", "--nb", "10", "--no-strict-rccl"]      # (no RCCL without GPUs: the statistic goes over the host group, and the line says so)
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    d = _check(lines[0], 2)
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "shard2"
    assert "cpu_baseline" not in d and "alt_precision" not in d and "steady_state" not in d        # rank-0, N=1 only
    # every rank's own clock, gathered over the host group: the line carries min / mean / max and the efficiency against the best rank
    pr = d["per_rank"]
    assert len(pr["ranks"]) == 2 and {"ms_per_step", "kernel_avg_ms", "steps_per_s"} <= set(pr["ranks"][0])
    for k in ("ms_per_step", "kernel_avg_ms", "steps_per_s"):
        assert pr[k]["min"] <= pr[k]["mean"] <= pr[k]["max"]
    assert abs(d["efficiency_vs_best_rank"] - d["value"] / (2 * pr["steps_per_s"]["max"])) < 1e-12 and d["efficiency_vs_best_rank"] <= 1.0 + 1e-9
    assert d["config"]["batch_per_gpu"] == 2 and d["config"]["rccl_ok"] is False and "gloo" in d["config"]["stats_reduce"]


def test_two_ranks_strict_rccl_is_the_default():
    """Under WORLD_SIZE > 1 a reduction that did not go through RCCL fails the run unless --no-strict-rccl says otherwise."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_bench_emu.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--preroll", "1",
           "--batch", "1", "--horizon", "5", "--nb", "0"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["config"]["rccl_ok"] is False      # the line is still printed, and says which reduction ran


def test_eight_ranks_reduce_both_statistics_through_the_rccl_branch():
    """Dry run of the driver's 8-GPU launch (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8): eight ranks on the
    lane emulator, with the RCCL branch of libmpcq ACTIVE -- MPCQ_RCCL_LIB points the library at a test-only stand-in that reduces over
    shared memory (tests/fake_rccl) -- and strict mode on (the default): unique id from rank 0, one communicator per rank, the configs[1]
    headline reduced through it, the configs[3] swarm leg (its own engine, here 2 quadrotors per rank) reduced through the SAME
    communicator (mpcq_comm_share), both statistics complete, the line self-consistent over 8 ranks."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_bench_emu.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--preroll", "2",
           "--batch", "1", "--horizon", "5", "--nb", "0", "--swarm-per-rank", "2"]
    env = dict(os.environ, MPCQ_RCCL_LIB=FAKE_RCCL, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout.decode()[-3000:]           # strict mode: a reduction that fell back to the host group would exit non-zero
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-3000:]
    d = _check(lines[0], 8)
    assert d["config"]["global_batch"] == 8 and d["config"]["batch_per_gpu"] == 1 and d["config"]["parallelism"] == "shard8"
    assert d["config"]["stats_reduce"] == "rccl" and d["config"]["rccl_ok"] is True
    pr = d["per_rank"]
    assert len(pr["ranks"]) == 8
    assert abs(d["efficiency_vs_best_rank"] - d["value"] / (8 * pr["steps_per_s"]["max"])) < 1e-12 and 0 < d["efficiency_vs_best_rank"] <= 1.0 + 1e-9
    # every rank reports how long its host-side generation of the references took under 8 concurrent ranks (round-5 verdict, item 9): the
    # driver's budget is minutes; here (tiny shapes, eight processes on the CPUs of this container) seconds
    assert all(0 <= r["host_generation_s"] < 120 for r in pr["ranks"])
    # the swarm leg: 2 quadrotors on each of 8 ranks, its statistic over the same communicator
    sw = d["swarm"]
    assert sw["n_gpus"] == 8 and sw["global_batch"] == 16 and sw["stats_reduce"] == "rccl"
    assert len(sw["per_rank"]["ranks"]) == 8 and 0 < sw["efficiency_vs_best_rank"] <= 1.0 + 1e-9
    assert sw["tracking_steps"] == 16 * (2 + 5 + 20)              # every rank's (pre-roll + warm-up + timed) periods arrived in the sum
    assert d["tracking"]["steps"] == 8 * (2 + 1 + 2)
