"""The CHECKED build of the product kernels (-DMPCQ_CHECKED: every pointer a fat pointer that checks its index against its
region -- LDS workspace, double block, per-instance global record, trajectory, state records -- and every cross-lane
operation checking that all 64 lanes take part) on the lane emulator, under UBSan, with the lanes resumed in forward and
in shuffled order.  CPU only: on the GPU the same checked build runs as libmpcq_checked.so
(tools/checked_gpu_suite.sh).  A violation makes the C call fail with a message naming the region, the index and the lane;
a lane missing from a cross-lane operation or a UBSan finding aborts the subprocess."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "wave_emu")
LIB = os.path.join(EMU_DIR, "libmpcq_emu_checked.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

CODE = """
import sys, json, numpy as np
sys.path[:0] = [%r, %r]
import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine
make = lambda cfg: Engine(cfg, lib_path=%r)
out = {}
out["teacher_forced_active_bounds"] = pc.case_teacher_forced_log(make, "log_trajectory_v15_a5_gp2.npz", 8)
worst, hist, failed = pc.case_saturating_references(make, B=2, K=12)
out["saturating"] = dict(worst=worst, failed=failed, fallbacks=sum(n for v, n in hist.items() if v >= 1000), multi_pass=sum(n for v, n in hist.items() if 2 <= v %% 1000))
out["swarm"] = pc.case_swarm_closed_loop(make, B=2, N=10, nb=10, K=5)
out["swarm_f32"] = pc.case_swarm_closed_loop(make, B=2, N=10, nb=10, K=4, precision=1)
pc.case_explicit_api(make, B=2, N=5, nb=10)
import dataclasses
make_c = lambda cfg: make(dataclasses.replace(cfg, tune=dict(stage_mem="compact")))   # gains in the global record, r0/lb/ub inside the union
worst_c, hist_c, failed_c = pc.case_saturating_references(make_c, B=2, K=8)
out["compact"] = dict(worst=worst_c, failed=failed_c, fallbacks=sum(n for v, n in hist_c.items() if v >= 1000), swarm=pc.case_swarm_closed_loop(make_c, B=2, N=10, nb=10, K=3))
import test_engine_edges as te
te._ragged_and_exhausted(%r)
print(json.dumps(out))
"""


@pytest.fixture(scope="module")
def checked_lib():
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang on this runner")
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-j4", "-C", EMU_DIR, "checked"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ubsan = subprocess.check_output([CLANG, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"]).decode().strip()
    if not os.path.exists(ubsan):
        pytest.skip("no shared UBSan runtime")
    return ubsan


@pytest.mark.parametrize("order", ["shuffle"])      # lanes resumed in a pseudo-random order that changes at every rendezvous (forward and reversed order: tests/test_emu_parity.py)
def test_checked_build_clean_on_emulator(checked_lib, order):
    env = dict(os.environ, LD_PRELOAD=checked_lib, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    if order == "reverse":
        env["MPCQ_EMU_REVERSE"] = "1"
    if order == "shuffle":
        env["MPCQ_EMU_SHUFFLE"] = "20261003"
    out = subprocess.run([sys.executable, "-c", CODE % (HERE, os.path.dirname(HERE), LIB, LIB)], env=env, capture_output=True, timeout=900)
    assert out.returncode == 0, (out.stdout.decode()[-1500:], out.stderr.decode()[-3000:])
    r = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert r["teacher_forced_active_bounds"] < 1e-8 and r["swarm"] < 1e-7 and r["saturating"]["worst"] < 1e-7
    assert r["saturating"]["failed"] == 0 and r["saturating"]["fallbacks"] > 0 and r["saturating"]["multi_pass"] > 0
    assert r["compact"]["worst"] < 1e-7 and r["compact"]["failed"] == 0 and r["compact"]["fallbacks"] > 0 and r["compact"]["swarm"] < 1e-7


def test_checked_build_reports_an_out_of_range_index(checked_lib):
    """The checker itself: a trajectory shorter than the engine was told (Tmax understated through the C ABI is refused by the
    library, so the violation is provoked below it, with a cursor beyond the trajectory) must come back as an error naming the
    trajectory region, not as a wild read."""
    code = """
import sys, numpy as np
sys.path[:0] = [%r, %r]
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd import _lib
from mpc_quad_ros_amd.params import EngineConfig
e = Engine(EngineConfig(batch=1, N=5), lib_path=%r)
assert b"CHECKED" in e.lib.mpcq_version()
traj = np.zeros((1, 40, 13)); traj[:, :, 3] = 1.0
e.set_trajectories(traj)
e.set_state(idx=np.array([-7]))          # a corrupted cursor: rows idx + j skip < 0 (the library refuses it unless told not to, see env below)
try:
    e.step(traj[:, 0].copy())
except _lib.MpcqError as ex:
    print("REPORTED", ex)
""" % (HERE, os.path.dirname(HERE), LIB)
    env = dict(os.environ, LD_PRELOAD=checked_lib, MPCQ_TUNING="1", MPCQ_SKIP_STATE_CHECKS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert "REPORTED" in out.stdout.decode() and "region tag 6" in out.stdout.decode(), out.stdout.decode()
