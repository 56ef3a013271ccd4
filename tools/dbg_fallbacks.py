#!/usr/bin/env python3
"""Diagnostic: per-step count of IPM fallbacks / failures of the closed-loop benchmark scenario."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories

prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
lib = sys.argv[2] if len(sys.argv) > 2 else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
B, N, nb = 1024, 20, 10
kw = dict(lib_path=os.path.join(ROOT, "mpc_quad_ros_amd", lib)) if lib else {}
e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=prec), **kw)
traj, lens = swarm_trajectories(2026, 0, B)
e.set_trajectories(traj, lens)
e.sim_reset(np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1)))
tot_fb = tot_bad = 0
for k in range(steps):
    e.sim_steps(1, 2, 5e-3)
    it = e.get_qp_iter(); st = e.get_status()
    fb = int((it >= 1000).sum()); bad = int((st != 0).sum())
    tot_fb += fb; tot_bad += bad
    if fb or bad:
        print(f"step {k}: fallbacks {fb} status!=0 {bad} statuses {np.unique(st)} iters(max) {it.max()}")
print(f"{'f32' if prec else 'f64'} lib={lib}: total fallbacks {tot_fb}, failures {tot_bad}, mean passes {np.mean(it % 1000):.3f}")
