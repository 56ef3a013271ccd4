"""Lockstep throughput of the BASELINE configurations in both precisions, same workload and pre-roll (the numbers behind the f32 legs
of bench.py).  usage: python tools/precision_legs.py [steps]   -> gpurun_out/precision_legs.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpc_quad_ros_amd.params import PRECISION_F32, PRECISION_F64  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [("configs[1]", 1024, 20, 10, 600), ("configs[3] per rank", 8192, 20, 10, 600), ("configs[2]", 8192, 20, 20, 300), ("configs[4]", 4096, 50, 50, 300)]
if os.environ.get("PL_ONLY"):
    SHAPES = [s for s in SHAPES if s[0] in os.environ["PL_ONLY"].split(";")]
refs = bench.workload(2026, 0, 8192, 600 + 5 + K)
out = []
for name, B, N, nb, pre in SHAPES:
    for prec in (PRECISION_F64, PRECISION_F32):
        r = bench.config_leg(name, refs, B, N, nb, prec, 0, pre, 5, K)
        out.append(r)
        print(json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/precision_legs.json", "w"), indent=1)
