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


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "wave_emu")], stdout=subprocess.DEVNULL)


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
                                   "--batch", "2", "--horizon", "5", "--nb", "10", "--no-cpu-baseline", "--no-alt"], cwd=ROOT)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 1)
    # the oracle-parity leg of the line (here: emulated lanes vs the oracle on the tiny workload)
    p = d["parity_on_workload"]
    assert p["quad_steps"] == 2 * 30 and p["max_rel_dev"] < 1e-7 and p["failed"] == 0
    assert d["roofline"]["traffic"] is None and "no PMC profile" in d["roofline"]["traffic_note"]


def test_two_ranks_under_torchrun():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_bench_emu.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--preroll", "3",
           "--batch", "2", "--horizon", "5", "--nb", "10"]
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    d = _check(lines[0], 2)
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "shard2"
    assert "cpu_baseline" not in d and "alt_precision" not in d        # rank-0, N=1 only
