"""What the slowest quadrotor of a lockstep launch is doing (GPU box): 200 periods of the bench workload, per launch the quadrotor with the most
factorisations: its qp_iter fields (passes + interior-point iterations, fallback, flip mark, why the warm attempt ended), work counters, launch time."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpc_quad_ros_amd.engine import qp_fallback, qp_flip, qp_passes, qp_warm_exit  # noqa: E402

B, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 200
refs = bench.workload(2026, 0, B, bench.PREROLL + K + 30)
e, _ = bench.make_engine(B, 20, 10, 0, 0, 0, 2026, refs=refs)
e.sim_run(bench.PREROLL, 2, 5e-3)
rows, kinds = [], collections.Counter()
for k in range(K):
    e.sim_steps(1, 2, 5e-3)
    kt, _ = e.get_kernel_time()
    it = e.get_qp_iter(); fac, swp = e.get_qp_work()
    q = int(np.argmax(fac * 1000 + swp))
    kind = (int(qp_fallback(it)[q]), int(qp_flip(it)[q]), int(qp_warm_exit(it)[q]))
    kinds[kind] += 1
    rows.append((1e3 * kt, int(fac[q]), int(swp[q]), kind, int(qp_fallback(it).sum())))
rows = np.array([(r[0], r[1], r[2], r[4]) for r in rows])
print(f"launch ms mean {rows[:,0].mean():.4f}; slowest quadrotor: factorisations mean {rows[:,1].mean():.2f} (min {rows[:,1].min():.0f}, max {rows[:,1].max():.0f}), sweeps mean {rows[:,2].mean():.1f}; fallbacks per launch mean {rows[:,3].mean():.2f}")
print("slowest quadrotor's (fallback, flip mark, warm-exit reason) -> launches:", dict(kinds))
for lo, hi in ((0, 2), (2, 4), (4, 8), (8, 10), (10, 12), (12, 16), (16, 99)):
    sel = (rows[:, 1] >= lo) & (rows[:, 1] < hi)
    if sel.any():
        print(f"  slowest has {lo:2d}..{hi - 1:2d} factorisations: {int(sel.sum()):3d} launches, mean launch {rows[sel, 0].mean():.4f} ms")
a = np.polyfit(rows[:, 1], rows[:, 0], 1)
print(f"launch ms ~ {a[1]:.4f} + {a[0]:.4f} x factorisations of the slowest quadrotor")
