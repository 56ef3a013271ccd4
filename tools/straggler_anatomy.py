"""What the slowest quadrotor of a lockstep launch executes (bench workload, fp64): interior-point iterations against active-set
factorisations, from the work counters (mpcq_get_qp_work: factorisations | sweeps << 16).  usage: straggler_anatomy.py [launches]"""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 1024
refs = bench.workload(2026, 0, B, 600 + K + 30)
e, _ = bench.make_engine(B, 20, 10, 0, 0, 0, 2026, refs=refs)
e.sim_run(600, 2, 5e-3)
rows = []
for k in range(K):
    e.sim_steps(1, 2, 5e-3)
    fac, swp = e.get_qp_work()
    it = e.get_qp_iter()
    cost = fac * 632 + swp * 73
    b = int(np.argmax(cost))
    fb = (it[b] // 1000) % 10 != 0
    ipm = max(0, (int(swp[b]) - int(fac[b]) - 2) // 2) if fb else 0      # sweeps - factorisations = 2 it + 2 (+1)
    rows.append((int(fac[b]), ipm, int(fac[b]) - ipm, int(fb), int((it[b] // 10000) % 10 != 0), int(it[b] // 100000)))
r = np.array(rows)
print(f"{K} launches: slowest quadrotor mean factorisations {r[:,0].mean():.2f} = interior-point iterations {r[:,1].mean():.2f} + active-set factorisations {r[:,2].mean():.2f}; "
      f"fallback in {r[:,3].mean()*100:.0f} % of the launches, flip-marked {r[:,4].mean()*100:.0f} %")
print("interior-point iterations of the slowest:", dict(sorted(collections.Counter(r[:,1].tolist()).items())))
print("active-set factorisations of the slowest:", dict(sorted(collections.Counter(r[:,2].tolist()).items())))
print("why its warm attempt ended (0 none / settled, 1 budget, 2 pins, 3 wrong, 4 bounce, 5 numeric, 6 skipped):", dict(sorted(collections.Counter(r[:,5].tolist()).items())))
