#!/usr/bin/env python3
"""Diagnostic (GPU box): teacher-forced relative control deviation of the fp32 QP mode against the fp64 oracle on the six
reference logs, step by step (the numbers behind the f32 rows of DESIGN.md section 5 and parity_cases.F32_LOG_BUDGET)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import config_for_log, load_golden
from mpc_quad_ros_amd.engine import Engine
from oracle.oracle import OracleEngine
import parity_cases as pc

LOGS = [("log_traj1_v10_a10_gp0.npz", 60), ("log_traj0_v10_a10_gp2.npz", 110), ("log_traj0_v15_a5_gp2.npz", 150),
        ("log_trajectory_v15_a5_gp2.npz", 80), ("log_traj2_v10_a10_gp2.npz", 100), ("log_traj1_v15_a5_gp2.npz", 45)]
out = {}
for name, K in LOGS:
    g = load_golden(name)
    e, o = Engine(config_for_log(g, precision=1)), OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None]); o.set_trajectories(g["x_ref"][None])
    errs, its, wmax = [], [], []
    for k in range(K):
        e.set_state(**o.get_state())
        w, _ = e.step(g["x_odom"][k][None]); wo, _ = o.step(g["x_odom"][k][None])
        errs.append(pc.rel_err(w, wo)); its.append(int(e.get_qp_iter()[0])); wmax.append(float(np.abs(wo).max()))
    errs = np.array(errs)
    over = [(int(k), float(errs[k]), its[k], wmax[k]) for k in np.nonzero(errs >= 1e-4)[0]]
    out[name] = dict(steps=K, worst=float(errs.max()), median=float(np.median(errs)), p99=float(np.quantile(errs, 0.99)), over_1e4=over)
    print(name, "worst %.2e median %.2e steps >= 1e-4: %d" % (errs.max(), np.median(errs), len(over)), [(k, "%.1e" % v, it) for k, v, it, _ in over])
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f32_log_report.json"), "w"), indent=1)
