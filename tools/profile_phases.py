#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle breakdown of the fused step kernel (needs libmpcq_prof.so,
`make -C mpc_quad_ros_amd/csrc prof`).  Never used for reported timings: stamps perturb the kernel."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine  # noqa: E402
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace  # noqa: E402
from mpc_quad_ros_amd.trajectories import swarm_trajectories  # noqa: E402

NAMES = ["load", "shoot_x", "shoot_s", "factor", "fwd", "bwd", "adjoint", "rollout", "update", "post", "total"]
FINE = {11: "fwd: loads/shift", 12: "fwd: K Dx chain", 13: "fwd: broadcast", 14: "fwd: A Dx chain"}
if os.environ.get("PROF_MODE") == "serial":   # library built with -DMPCQ_PROFILE_SERIAL (PROF_LIB names it): the single-lane blocks and the parts of the post phase
    FINE = {11: "load: plant RK4 (lane 0)", 12: "post: nominal RK4 (lane 0)", 13: "post: drag, stats (lane 0)", 14: "post: RGP regress"}
if os.environ.get("PROF_MODE") == "other":   # library built with PROF_EXTRA=-DMPCQ_PROFILE_OTHER (PROF_LIB names it): what lies between the bracketed phases (fp64)
    FINE = {11: "other: shooting -> QP", 12: "other: pass set-up", 13: "other: factor -> sweep", 14: "other: ratio test, multipliers, step", 15: "other: QP -> full step"}
if os.environ.get("PROF_MODE") == "fac":   # library built with PROF_EXTRA=-DMPCQ_PROFILE_FAC: slots 11..15 split the factorisation
    FINE = {11: "fac: tile products", 12: "fac: LDS hand-over", 13: "fac: 4x4 LDL^T", 14: "fac: solves + stores", 15: "fac: P update + loop"}


def main():
    prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
    B, N, nb = 1024, 20, 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    lib = os.path.join(ROOT, "mpc_quad_ros_amd", "libmpcq_prof_fac.so" if os.environ.get("PROF_MODE") == "fac" else "libmpcq_prof.so")
    lib = os.environ.get("PROF_LIB", lib)
    import bench
    refs = bench.workload(2026, 0, B, int(os.environ.get('PREROLL', bench.PREROLL)) + steps)
    e, _ = bench.make_engine(B, N, nb, prec, 0, 0, 2026, lib_path=lib, refs=refs)      # the bench workload (min-snap references)
    e.sim_run(int(os.environ.get('PREROLL', bench.PREROLL)), 2, 5e-3)   # same regime as bench.py
    acc = np.zeros((B, 16))
    mx = np.zeros(16)
    its = []
    for k in range(steps):
        e.sim_steps(int(os.environ.get('PROF_K', 1)), 2, 5e-3)   # PROF_K=2: the profiled (last) launch carries the plant update at its head
        out = np.zeros((B, 16), dtype=np.uint64)
        rc = e.lib.mpcq_debug_profile(e.h, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        acc += out
        worst = out[:, 10].argmax()
        mx += out[worst]
        its.append(e.get_qp_iter() % 10000)
    its = np.array(its)
    mean = acc.mean(axis=0) / steps
    mxs = mx / steps
    print(f"precision {'f32' if prec else 'f64'}  B={B} N={N} nb={nb}  steps={steps}  passes mean {its.mean():.2f} max {its.max()}")
    print(f"{'phase':10s} {'mean cycles':>12s} {'share':>7s} {'slowest-quad cycles':>20s}")
    for i, n in enumerate(NAMES):
        print(f"{n:10s} {mean[i]:12.0f} {100 * mean[i] / mean[10]:6.1f}% {mxs[i]:20.0f}")
    for k, n in FINE.items():
        if mean[k] > 0:
            print(f"  {n:22s} {mean[k]:12.0f} {mxs[k]:20.0f}")
    if mean[14] > 0 and os.environ.get("PROF_MODE") not in ("fac", "serial", "other"):   # active-set statistics of the fp64 path (diagnostic counters of polish())
        tot = acc.sum(axis=0)
        print(f"factorisations per quad-step {mean[14]:.3f} (slowest quad of a launch: {mxs[14]:.2f}); stages visited per factorisation {tot[13] / tot[14]:.1f}; "
              f"pins per quad-step {mean[11]:.3f}, passes with a release {mean[12]:.3f}, with both {mean[15]:.3f}")
    other = mean[10] - mean[:10].sum()
    print(f"{'other':10s} {other:12.0f} {100 * other / mean[10]:6.1f}%")


if __name__ == "__main__":
    main()
