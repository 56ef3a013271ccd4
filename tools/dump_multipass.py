#!/usr/bin/env python3
"""Diagnostic (GPU box): collects bench-workload quadrotor-steps whose QP needed many factorisations, as replayable cases
(state before the step, measurement, trajectory) for the lane emulator (tools/replay_multipass.py, CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/multipass_cases.npz"
min_passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, N, nb = 1024, 20, 10
import bench
e, _ = bench.make_engine(B, N, nb, 0, 0, 0, 2026)
e.sim_steps(bench.PREROLL, 2, 5e-3)
cases = []
for k in range(120):
    st, sol, x = e.get_state(), e.get_solver_state(), e.sim_get_state()[0]
    e.sim_steps(1, 2, 5e-3)
    it = e.get_qp_iter()
    sel = (it >= 1000) if min_passes >= 1000 else ((it % 1000) >= min_passes)     # min_passes >= 1000: the solves that fell back
    for b in np.nonzero(sel)[0][:3]:
        if len(cases) < 60:
            cases.append(dict(b=b, step=bench.PREROLL + k, passes=it[b], x=x[b], X=st["X"][b], U=st["U"][b], mu=st["mu"][b], C=st["C"][b],
                              xpp=st["x_pred_prev"][b], hp=st["has_prev"][b], idx=st["idx"][b], qp_iter=sol["qp_iter"][b], w=e.sim_get_state()[1][b]))
keys = cases[0].keys()
np.savez_compressed(out, traj_idx=np.array([c["b"] for c in cases]), **{k: np.array([c[k] for c in cases]) for k in keys})
print("saved", len(cases), "cases; passes", [int(c["passes"]) for c in cases])
