#!/usr/bin/env python3
"""Diagnostic (GPU box): closed-loop run of the bench workload in fp64 with every solve's status read; prints the solves that fail or whose
float interior point broke down (mpcq_get_qp_work bit 15), and saves the engine state in front of the first one for a replay.
usage: [MPCQ_LIB=...] hybrid_failure_probe.py seed periods [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
seed, K = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
refs = bench.workload(seed, 0, B, K + 10)
e, _ = bench.make_engine(B, 20, 10, 0, 0, 0, seed, periods=K + 10, refs=refs)
nbad = nbrk = 0
for k in range(K):
    e.sim_steps(1, 2, 5e-3)
    st = e.get_status(); it = e.get_qp_iter(); brk = e.get_qp_float_breakdown(); fac, swp = e.get_qp_work()
    for b in np.nonzero(((st & 7) != 0) | brk)[0]:
        print(f"period {k} quadrotor {b}: status {st[b]} qp_iter {it[b]} factorisations {fac[b]} sweeps {swp[b]} float breakdown {bool(brk[b])}", flush=True)
    nbad += int(((st & 7) != 0).sum()); nbrk += int(brk.sum())
print(f"seed {seed}: {K} periods x {B}: failed solves {nbad}, float breakdowns {nbrk}, library {e.lib.mpcq_version().decode()}")
