"""Solver heuristics against the lockstep rate of the bench workload (B quadrotors behind the 600-period pre-roll, K periods timed):
one engine per setting of mpcq_tuning, same references.  usage: python tools/tune_sweep.py [B] [K] [seed]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpc_quad_ros_amd.engine import Engine, qp_fallback
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2026
refs = bench.workload(seed, 0, B, 600 + K + 30)
SETTINGS = [{}, {"ipm_tol": 3e-6}, {"ipm_tol": 1e-5}, {"ipm_tol": 1e-4}, {"ipm_tol": 1e-7},
            {"pin_ratio": 0.1}, {"pin_ratio": 0.4}, {"pin_ratio": 1.0},
            {"warm_max": 4}, {"warm_max": 8}, {"warm_max": 12}, {"warm_retry": 2}, {"warm_retry": 3},
            {"flip_max": 1}, {"flip_max": 4}, {"flip_max": 8}, {"flip_max": -1},
            {"abort_pins": 6}, {"abort_pins": 16}, {"abort_pins": -1}, {"abort_wrong": 5}, {"abort_wrong": 14}, {"abort_wrong": -1},
            {"ipm_mu0": 3e-5}, {"ipm_mu0": 3e-4}, {"ipm_margin": 0.05}, {"ipm_margin": 0.2}, {}]
if os.environ.get("SWEEP"):
    SETTINGS = json.loads(os.environ["SWEEP"])
for tune in SETTINGS:
    e = Engine(EngineConfig(batch=B, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10), tune=tune or None,
                            precision=1 if os.environ.get("SWEEP_F32") else 0))
    e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0, (B, 1)))
    e.sim_run(600, 2, 5e-3)
    e.sim_steps(5, 2, 5e-3); e.synchronize()
    t0 = time.perf_counter()
    fb = 0
    for k in range(K // 50):
        e.sim_steps(50, 2, 5e-3)
    e.synchronize()
    dt = time.perf_counter() - t0
    its = e.get_qp_iter()
    st = e.get_status()
    t = e.get_tracking_stats()
    print(json.dumps({"tune": tune, "steps_per_s": round(B * (K // 50) * 50 / dt), "fallbacks_last": int(qp_fallback(its).sum()), "failed_last": int(((st & 7) != 0).sum()),
                      "rms_pos": round(float(np.sqrt(t[0] / (3 * max(t[2], 1)))), 5)}), flush=True)
    e.close()
