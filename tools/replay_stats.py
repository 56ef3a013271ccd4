#!/usr/bin/env python3
"""Diagnostic (CPU): replays the quadrotor-steps collected by tools/dump_multipass.py through the lane emulator under several
settings of the working-set heuristics (environment knobs of mpcq_create) and prints the factorisation counts side by side.
usage: replay_stats.py cases.npz "NAME|K=V,K=V" ...   (the minimiser does not depend on the sequence of working sets, so the
controls agree to rounding in every column; the script checks that)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
import bench

subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "wave_emu")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
lib = os.path.join(ROOT, "tests", "wave_emu", "libmpcq_emu.so")
d = np.load(sys.argv[1])
variants = [a.split("|") for a in sys.argv[2:]] or [["default", ""]]
ncase = int(os.environ.get("NCASE", len(d["b"])))
N, nb = 20, 10
res = {v[0]: [] for v in variants}
for c in range(ncase):
    b = int(d["b"][c])
    traj, lens = bench.workload(2026, b, 1, 1000)
    row, ws = [], []
    for name, envs in variants:
        kv = dict(e.split("=") for e in envs.split(",") if e)
        os.environ.update(kv)
        e = Engine(EngineConfig(batch=1, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb)), lib_path=lib)
        for k in kv: del os.environ[k]
        e.set_trajectories(traj, lens)
        e.set_state(X=d["X"][c][None], U=d["U"][c][None], mu=d["mu"][c][None], C=d["C"][c][None], x_pred_prev=d["xpp"][c][None],
                    has_prev=d["hp"][c:c + 1], idx=d["idx"][c:c + 1])
        e.set_solver_state(qp_iter=d["qp_iter"][c:c + 1])
        w, _ = e.step(d["x"][c][None])
        it = int(e.get_qp_iter()[0]); res[name].append(it); row.append(it); ws.append(w[0])
        e.close()
    dev = max(np.abs(w - ws[0]).max() for w in ws)
    print(f"case {c:2d} quad {b:4d} gpu {int(d['passes'][c]):5d} | " + " ".join(f"{n}={i}" for (n, _), i in zip(variants, row)) + f" | dw {dev:.1e}", flush=True)
for name, _ in variants:
    a = np.array(res[name]); fb = a >= 1000
    print(f"{name:12s} fallbacks {fb.sum():3d}/{len(a)}  mean passes of the settled {a[~fb].mean() if (~fb).any() else 0:.2f}  max {a[~fb].max() if (~fb).any() else 0}  mean cost of fallbacks {(a[fb] - 1000).mean() if fb.any() else 0:.2f}")
