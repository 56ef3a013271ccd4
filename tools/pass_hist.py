#!/usr/bin/env python3
"""Diagnostic: histogram of QP passes per quad-step on the bench workload (uses libmpcq.so)."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 220
B, N, nb = 1024, 20, 10
import bench
refs = bench.workload(2026, 0, B, int(os.environ.get('PREROLL', bench.PREROLL)) + steps)
e, _ = bench.make_engine(B, N, nb, prec, 0, 0, 2026, refs=refs)        # the bench workload (min-snap references)
e.sim_run(int(os.environ.get('PREROLL', bench.PREROLL)), 2, 5e-3)   # same regime as bench.py
hist = collections.Counter(); permax = []; kt = []
for k in range(steps):
    e.sim_steps(1, 2, 5e-3)
    it = e.get_qp_iter() % 10000      # passes + 1000 x fallback (flip mark and warm-exit reason dropped)
    hist.update(it.tolist()); permax.append(int(it.max())); kt.append(e.get_kernel_time()[0])
    if k > 0 and (it >= 1000).any() and len(sys.argv) > 3: print('fallback step', k, 'quads', np.nonzero(it >= 1000)[0].tolist(), it[it >= 1000].tolist())
    if ((e.get_status() & 7) != 0).any(): print("step", k, "failed instances", np.nonzero(e.get_status())[0], e.get_status()[e.get_status() != 0], it[e.get_status() != 0])
tot = sum(hist.values())
print("passes histogram (value: share):", {k: round(v / tot, 5) for k, v in sorted(hist.items())})
fb = sum(v for k, v in hist.items() if k >= 1000)
print("fallback share", fb / tot, "steps with >=1 fallback", sum(1 for m in permax if m >= 1000), "of", steps)
kt = np.array(kt) * 1e3
print("kernel ms: mean %.3f  with-fallback mean %.3f  no-fallback mean %.3f" % (kt.mean(), kt[np.array(permax) >= 1000].mean() if any(m >= 1000 for m in permax) else 0, kt[np.array(permax) < 1000].mean()))
pm = np.array(permax) % 1000
for v in sorted(set(pm.tolist())):
    sel = (pm == v) & (np.array(permax) < 1000)
    if sel.any(): print("launches whose slowest instance took %d passes: %d, kernel ms mean %.4f" % (v, sel.sum(), kt[sel].mean()))
