"""Two builds of the library on the same closed-loop run: are the results bit-identical (if not: how far apart after the first period, where
only rounding can differ, and at the end), and which is faster?
Each build runs in its own process (two libmpcq in one process would share symbols).
usage: python tools/compare_builds.py libA.so libB.so [B N nb preroll steps [f64|f32] [stage_mem]]"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib, dump, B, N, nb, pre, K, prec, stage_mem):
    import bench
    from mpc_quad_ros_amd.engine import Engine
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
    tune = dict(stage_mem=stage_mem) if stage_mem else None
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=prec, tune=tune), lib_path=lib)
    refs = bench.workload(2026, 0, B, pre + K + 30)
    e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0, (B, 1)))
    e.sim_steps(1, 2, 5e-3)
    x1, w1 = e.sim_get_state()
    first = dict(x1=x1.copy(), w1=w1.copy(), it1=e.get_qp_iter().copy())
    e.sim_steps(pre - 1, 2, 5e-3)
    e.synchronize()
    t0 = time.perf_counter()
    e.sim_steps(K, 2, 5e-3)
    e.synchronize()
    dt = time.perf_counter() - t0
    kt, kl = e.get_kernel_time()
    x, w = e.sim_get_state()
    h = hashlib.sha256()
    for a in (x, w, e.get_qp_iter(), e.get_status(), e.get_rgp()[0]):
        h.update(np.ascontiguousarray(a).tobytes())
    np.savez(dump, x=x, w=w, it=e.get_qp_iter(), **first)
    print(json.dumps({"lib": os.path.basename(lib), "steps_per_s": B * K / dt, "kernel_avg_ms": 1e3 * kt / max(kl, 1), "sha256": h.hexdigest()}))


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], sys.argv[3], *(int(v) for v in sys.argv[4:11]))
        sys.exit(0)
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    a = sys.argv[3:]
    B, N, nb, pre, K = (int(v) for v in (a[:5] + ["1024", "20", "10", "200", "100"][len(a[:5]):]))
    prec = 1 if len(a) > 5 and a[5] == "f32" else 0
    sm = int(a[6]) if len(a) > 6 else 0
    res = []
    for k, lib in enumerate(libs):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib, f"/tmp/compare_builds_{k}.npz"] + [str(v) for v in (B, N, nb, pre, K, prec, sm)], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-2000:]); sys.exit(1)
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    da, db = np.load("/tmp/compare_builds_0.npz"), np.load("/tmp/compare_builds_1.npz")
    diff = {k: float(np.max(np.abs(da[k].astype(float) - db[k].astype(float)))) for k in ("x1", "w1", "it1", "x", "w", "it")}
    print(json.dumps({"max_abs_difference": diff, "B": B, "N": N, "nb": nb, "preroll": pre, "steps": K, "precision": "f32" if prec else "f64", "stage_mem": sm,
                      "bit_identical": res[0]["sha256"] == res[1]["sha256"], "speedup_b_over_a": res[1]["steps_per_s"] / res[0]["steps_per_s"], "runs": res}))
