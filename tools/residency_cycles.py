"""Diagnostic (GPU box, libmpcq_prof.so): what co-residency costs a wavefront, phase by phase.  The compact-layout lockstep launch at
B = 256 k (k = 1 .. 6 workgroups per CU in fp64, .. 8 in f32; one round of workgroups, all resident together), cycle stamps of the SAME
first 256 quadrotors of the bench workload every time (their arithmetic is identical in every run: results do not depend on the batch).
usage: python tools/residency_cycles.py [f64|f32] [launches]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpc_quad_ros_amd.engine import Engine  # noqa: E402
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace  # noqa: E402

NAMES = ["load", "shoot_x", "shoot_s", "factor", "fwd", "bwd", "adjoint", "rollout", "update", "post", "total"]
prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pre = 600
ks = [1, 2, 3, 4, 5, 6] + ([7, 8] if prec else [])
lib = os.environ.get("PROF_LIB", os.path.join(ROOT, "mpc_quad_ros_amd", "libmpcq_prof.so"))
traj, lens = bench.workload(2026, 0, 256 * ks[-1], pre + steps + 30)
rows, digests = [], []
for k in ks:
    B = 256 * k
    e = Engine(EngineConfig(batch=B, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10), precision=prec,
                            tune=dict(stage_mem=3, groups=1, block_order=1)), lib_path=lib)
    e.set_trajectories(traj[:B], lens[:B]); e.sim_reset(np.tile(bench.X0, (B, 1)))
    e.sim_run(pre, 2, 5e-3)
    acc = np.zeros(16)
    for _ in range(steps):
        e.sim_steps(1, 2, 5e-3)
        out = np.zeros((B, 16), dtype=np.uint64)
        assert e.lib.mpcq_debug_profile(e.h, out.ctypes.data_as(ctypes.c_void_p)) == 0
        acc += out[:256].astype(np.float64).mean(axis=0)
    x, w = e.sim_get_state()
    digests.append(float(np.sum(x[:256]) + np.sum(w[:256])))
    rows.append(acc / steps)
    e.close()
print(f"precision {'f32' if prec else 'f64'}  compact layout, N=20 nb=10, {steps} lockstep launches behind {pre} periods; mean shader cycles per step of quadrotors 0..255")
print(f"same results in every run: {all(d == digests[0] for d in digests)}")
print("per CU  " + " ".join(f"{n:>9s}" for n in NAMES) + "   total vs k=1   steps per CU per M cycles")
for k, r in zip(ks, rows):
    print(f"{k:6d}  " + " ".join(f"{r[i]:9.0f}" for i in range(11)) + f"   {r[10] / rows[0][10]:13.3f}   {1e6 * k / r[10]:10.2f}")
print("phase cycles relative to k=1:")
for k, r in zip(ks, rows):
    print(f"{k:6d}  " + " ".join(f"{(r[i] / rows[0][i] if rows[0][i] > 50 else float('nan')):9.3f}" for i in range(11)))
