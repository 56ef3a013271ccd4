"""Diagnostic (GPU box): lockstep launches against free-running launches of the same engine configuration in fp32, period by period:
when do the two first differ, and by how much?   usage: f32_mode_diff.py B N nb [periods]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B, N, nb = (int(v) for v in sys.argv[1:4]); K = int(sys.argv[4]) if len(sys.argv) > 4 else 40
refs = bench.workload(7, 0, B, K + 10)
a, _ = bench.make_engine(B, N, nb, 1, 0, 0, 7, refs=refs)
b, _ = bench.make_engine(B, N, nb, 1, 0, 0, 7, refs=refs)
first = None
for k in range(K):
    a.sim_steps(1, 2, 5e-3); b.sim_run(1, 2, 5e-3)
    sa, sb = a.get_state(), b.get_state()
    d = {nm: float(np.abs(sa[nm] - sb[nm]).max()) for nm in ("X", "U", "mu", "C")}
    nq = int((np.abs(sa["U"] - sb["U"]).reshape(B, -1).max(axis=1) > 0).sum())
    if any(v > 0 for v in d.values()) and first is None:
        first = k
        ita, itb = a.get_qp_iter(), b.get_qp_iter()
        q = int(np.argmax(np.abs(sa["U"] - sb["U"]).reshape(B, -1).max(axis=1)))
        print(f"first difference after period {k + 1}: {d}; quadrotors with different U: {nq}; worst quadrotor {q}: qp_iter lockstep {ita[q]} free-running {itb[q]}, status {a.get_status()[q]} / {b.get_status()[q]}")
print(f"after {K} periods: max |diff| {d}, quadrotors with different U {nq}; first difference at period {first}")
