"""Histogram of qp_iter fields (passes, fallback, flip mark, why the warm attempt ended) over lockstep periods of the bench workload.
usage: qp_iter_hist.py [f64|f32] [periods]"""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
prec = 1 if len(sys.argv) > 1 and sys.argv[1] == "f32" else 0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 1024
refs = bench.workload(2026, 0, B, 600 + K + 30)
e, _ = bench.make_engine(B, 20, 10, prec, 0, 0, 2026, refs=refs)
e.sim_run(600, 2, 5e-3)
why = collections.Counter(); passes = collections.Counter(); fb = 0; flip = 0
for k in range(K):
    e.sim_steps(1, 2, 5e-3)
    it = e.get_qp_iter()
    for v in it:
        why[int(v) // 100000] += 1
        passes[min(int(v) % 1000, 30)] += 1
    fb += int(((it // 1000) % 10 != 0).sum()); flip += int(((it // 10000) % 10 != 0).sum())
print("precision", "f32" if prec else "f64", "quad-steps", B * K, "fallback", fb, "flip-marked", flip)
print("why (0 none, 1 budget, 2 pins, 3 wrong, 4 bounce, 5 numeric, 6 skipped):", dict(sorted(why.items())))
print("passes + ipm iterations:", dict(sorted(passes.items())))
