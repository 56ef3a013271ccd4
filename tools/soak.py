#!/usr/bin/env python3
"""Diagnostic (GPU box): long closed-loop run of the bench workload in both precisions -- solver failures, tracking, and the
bitwise agreement of the lockstep and the free-running launch modes.  usage: [SOAK_B= SOAK_N= SOAK_NB= SOAK_EVERY=] soak.py [periods] [seed]
SOAK_EVERY: status / fallback sampling interval in periods (default 100; 1 = every solve is looked at)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B, N, nb = int(os.environ.get('SOAK_B', 1024)), int(os.environ.get('SOAK_N', 20)), int(os.environ.get('SOAK_NB', 10))
refs = bench.workload(seed, 0, B, K + 10)
for prec, name in [pn for pn in ((0, "f64"), (1, "f32")) if os.environ.get("SOAK_PREC", pn[1]) == pn[1]]:   # SOAK_PREC=f64 | f32: that precision only
    e1, _ = bench.make_engine(B, N, nb, prec, 0, 0, seed, periods=K + 10, refs=refs)
    e2, _ = bench.make_engine(B, N, nb, prec, 0, 0, seed, periods=K + 10, refs=refs)
    bad = 0; fb = 0; low = 0; brk = 0
    EV = int(os.environ.get("SOAK_EVERY", 100))
    for k0 in range(0, K, EV):
        e1.sim_steps(EV, 2, 5e-3)
        st = e1.get_status(); bad += int(((st & 7) != 0).sum()); low += int((st == 8).sum()); fb += int((e1.get_qp_iter() >= 1000).sum()); brk += int(e1.get_qp_float_breakdown().sum())
    e2.sim_run(K, 2, 5e-3)
    s1, s2 = e1.get_state(), e2.get_state()
    same = all(np.array_equal(s1[k], s2[k]) for k in ("X", "U", "mu", "C", "idx"))
    t = e1.get_tracking_stats()
    print(f"{name}: {K} periods x {B} quadrotors (N = {N}, nb = {nb}); at the {K // EV} sampled periods: failed solves {bad}, MPCQ_SOLVE_LOW_ACCURACY {low}, fallbacks {fb} (float interior point broken down in {brk}); "
          f"rms pos {np.sqrt(t[0] / (3 * max(t[2], 1))):.4f} m, max {np.sqrt(t[3]):.3f} m, failed instances {t[4]:.0f}; "
          f"lockstep == free-running bitwise: {same}", flush=True)
    e1.close(); e2.close()
