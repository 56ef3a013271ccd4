#!/usr/bin/env python3
"""Diagnostic: run the four step-kernel variants of one precision (stage records in LDS / global memory x shape-specialised /
any-shape instance) on the device and print how far their controls are apart.  usage: check_variants.py <0|1> <lib.so>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
B, N, nb = 8, 20, 10
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
traj, lens = swarm_trajectories(11, 0, B)
x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
for mem in ("lds", "global"):
    for generic in (False, True):
        os.environ["MPCQ_STAGE_MEM"] = mem
        if generic: os.environ["MPCQ_GENERIC"] = "1"
        else: os.environ.pop("MPCQ_GENERIC", None)
        e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=prec), lib_path=os.path.join(ROOT, "mpc_quad_ros_amd", sys.argv[2]))
        e.set_trajectories(traj, lens); e.sim_reset(x0)
        out = []
        for k in range(8):
            e.sim_steps(1, 2, 5e-3)
            out.append((e.get_status()[:3].tolist(), e.get_qp_iter()[:3].tolist(), float(e.sim_get_state()[1][0, 0])))
        ws = np.array([o[2] for o in out]); ref = ws if mem == "lds" and not generic else ref
        print(mem, "generic" if generic else "special", "status", [o[0][0] for o in out], "iters", [o[1][0] for o in out], "max |w - w_ref|", float(np.abs(ws - ref).max()))
