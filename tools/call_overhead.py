"""Fixed cost of one mpcq_sim_steps call (host side + first launch + final plant launch + synchronise) against the per-period cost:
wall time of calls of K periods, K = 1 ... 100, on the bench workload behind its pre-roll.  usage: python tools/call_overhead.py [B]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
refs = bench.workload(2026, 0, B, 1200)
e, _ = bench.make_engine(B, 20, 10, 0, 0, 0, 2026, refs=refs)
e.sim_run(600, 2, 5e-3)
e.sim_steps(5, 2, 5e-3); e.synchronize()
out = []
for K in (1, 1, 2, 5, 20, 20, 20, 100):
    e.synchronize()
    t0 = time.perf_counter()
    e.sim_steps(K, 2, 5e-3)
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    kt, kl = e.get_kernel_time()
    out.append({"K": K, "call_ms": 1e3 * (t1 - t0), "sync_ms": 1e3 * (t2 - t1), "ms_per_period": 1e3 * (t2 - t0) / K, "kernel_avg_ms_sampled": 1e3 * kt / max(kl, 1), "launches_sampled": kl})
print(json.dumps(out, indent=1))
