#!/usr/bin/env python3
"""Diagnostic (GPU box): lockstep throughput on the ROUND-1 bench workload (spline flights from hover, and 150 periods in), for
continuity of the numbers across rounds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
B, N, nb = 1024, 20, 10
X0 = np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
for prec in (0, 1):
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=prec))
    traj, lens = swarm_trajectories(2026, 0, B)
    e.set_trajectories(traj, lens); e.sim_reset(np.tile(X0, (B, 1)))
    for pre, K in ((0, 20), (150, 200)):
        if pre: e.sim_steps(pre - 25, 2, 5e-3)
        e.sim_steps(5, 2, 5e-3); e.synchronize()
        t0 = time.perf_counter(); e.sim_steps(K, 2, 5e-3); e.synchronize(); t1 = time.perf_counter()
        print(f"precision {prec}: round-1 workload (spline flights), {pre} periods in, K={K}: {B*K/(t1-t0)/1e6:.2f} M steps/s, {1e3*(t1-t0)/K:.4f} ms per period, passes mean {(e.get_qp_iter()%1000).mean():.2f}", flush=True)
    e.close()
