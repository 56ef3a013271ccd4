#!/usr/bin/env python3
"""Fixture of the three solves the widened f32 audit of round 6 turned up (profiles/r6_f32_audit_more.txt): the state of ONE quadrotor of the
bench workload just before the period in question, as tools/f32_audit.py saved it on the GPU box (gpurun_out/f32_audit_hit_<seed>_<N>_<n>.npz),
with the controls the fp64 and the mixed-precision engine returned there.  Two of the quadrotors are lost (QP gradient scale 1e9, 78 of 80
inputs at a bound): the f32 mode is outside its validity limit and has to say so; the third idles (largest control 0.008) and was returned
1.3e-6 of full thrust off with status 0.  Nothing of the reference is involved: inputs and expected outputs are this repo's own engines'.
usage: python tests/golden/make_lost_quadrotors.py gpurun_out/f32_audit_hit_5_20_0.npz gpurun_out/f32_audit_hit_4_20_0.npz gpurun_out/f32_audit_hit_4_20_1.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T_HORIZON = 1.0
out = {"cases": np.int32(len(sys.argv) - 1)}
for c, path in enumerate(sys.argv[1:]):
    d = np.load(path)
    N, idx, ln = int(d["N"]), int(d["st_idx"]), int(d["len"])
    skip = int((T_HORIZON / N) / 0.01)
    keep = min(ln - idx, N * skip + 2)       # the rows the chunk of this period reads (get_reference_chunk: N rows `skip` apart) -- same `have`
    p = f"c{c}_"
    out[p + "traj"] = d["traj"][idx:idx + keep]
    out[p + "len"] = np.int32(keep)
    for k in ("N", "nb", "x", "prev", "w32", "w64"):
        out[p + k] = d[k]
    for k in d.files:
        if k.startswith("st_"):
            out[p + k] = d[k] if k != "st_idx" else np.int32(0)
    out[p + "origin"] = np.array(f"{os.path.basename(path)}: period {int(d['k'])}, quadrotor {int(d['b'])}")
np.savez_compressed(os.path.join(HERE, "f32_lost_quadrotors.npz"), **out)
print("wrote", os.path.join(HERE, "f32_lost_quadrotors.npz"), os.path.getsize(os.path.join(HERE, "f32_lost_quadrotors.npz")), "bytes")
