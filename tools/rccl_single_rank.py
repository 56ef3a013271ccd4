#!/usr/bin/env python3
"""Diagnostic (GPU box): the one collective of the path on the real RCCL with a one-rank communicator (what a single GPU allows):
mpcq_comm_unique_id -> mpcq_comm_init(0 of 1) -> mpcq_allreduce_tracking_stats, result == the local statistic.  Run it under
`rocprofv3 --kernel-trace --memory-copy-trace --stats -- python3 tools/rccl_single_rank.py` to see what RCCL launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

e, _ = bench.make_engine(256, 20, 10, 0, 0, 0, 2026, periods=100)
e.sim_steps(30, 2, 5e-3)
local = e.get_tracking_stats()
uid = e.comm_unique_id()
e.comm_init(0, 1, uid)
for _ in range(3):
    red = e.allreduce_tracking_stats()
assert np.array_equal(red, local), (red, local)
print("RCCL one-rank all-reduce of the tracking statistic:", red.tolist(), "== local statistic")
e.close()
