"""Large-batch lockstep throughput (BASELINE configs[2] / configs[3] per rank): the launch order on and off, launch times.
usage: python tools/large_batch.py [B] [N] [nb] [preroll] [steps]   (environment: MPCQ_LIB for experiment builds)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpc_quad_ros_amd.engine import Engine, qp_fallback, qp_passes  # noqa: E402
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace  # noqa: E402

B, N, nb, pre, K = (int(v) for v in (sys.argv[1:6] + ["8192", "20", "10", "600", "40"][len(sys.argv) - 1:]))
refs = bench.workload(2026, 0, B, pre + K + 30)
out = {"B": B, "N": N, "nb": nb, "preroll": pre, "steps": K, "runs": []}
# (one group = one launch per period over the whole batch, as until round 5: the counter passes attribute their numbers per launch)
VARIANTS = {"identity": ("identity order", dict(block_order=1, groups=1)), "sorted": ("cost-sorted order", dict(block_order=2, groups=1)),
            "global": ("layout: stage records in global memory, cost-sorted order", dict(stage_mem=2, groups=1)),
            "compact": ("layout: compact (gains in global memory, 256 registers), cost-sorted order", dict(stage_mem=3, groups=1))}
VARIANTS["global_g2"] = ("layout: stage records in global memory (4 per CU in f64), two groups", dict(stage_mem=2, groups=2))
VARIANTS["compact_g2"] = ("layout: compact, two groups", dict(stage_mem=3, groups=2))
for g in (1, 2, 3, 4, 6, 8, 16):      # mpcq_tuning.groups: the batch as g groups, each in lockstep on its own stream (automatic layout and order)
    VARIANTS[f"g{g}"] = (f"{g} group(s)", dict(groups=g))
which = os.environ.get("LB_VARIANTS", "identity,sorted").split(",")
for name, tune in (VARIANTS[w] for w in which):
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), tune=tune, precision=1 if os.environ.get("LB_F32") else 0))
    e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0, (B, 1)))
    if os.environ.get("LB_PREROLL_LOCKSTEP"):      # (profiler runs: no long persistent launch under counter collection)
        e.sim_steps(pre, 2, 5e-3)
    else:
        e.sim_run(pre, 2, 5e-3)
    e.sim_steps(5, 2, 5e-3)
    e.synchronize()
    t0 = time.perf_counter()
    e.sim_steps(K, 2, 5e-3)
    e.synchronize()
    dt = time.perf_counter() - t0
    kt, kl = e.get_kernel_time()
    kmin, kmax = e.get_kernel_time_minmax()
    its = e.get_qp_iter()
    x, w = e.sim_get_state()
    out["runs"].append({"order": name, "steps_per_s": B * K / dt, "ms_per_period": 1e3 * dt / K, "kernel_avg_ms": 1e3 * kt / max(kl, 1),
                        "kernel_min_ms": 1e3 * kmin, "kernel_max_ms": 1e3 * kmax, "mean_passes": float(qp_passes(its).mean()),
                        "fallbacks_last": int(qp_fallback(its).sum()), "digest": float(np.sum(x) + np.sum(w))})
    e.close()
out["bitwise_equal"] = all(r["digest"] == out["runs"][0]["digest"] for r in out["runs"])
print(json.dumps(out))
