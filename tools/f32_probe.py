"""Diagnostic: teacher-forced relative control deviation of the f32 (mixed-precision) mode against the fp64 oracle on the swarm workload,
per quadrotor and step, with status and qp_iter.  usage: f32_probe.py B N nb K start   (LIB=... selects the library)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
from oracle.oracle import OracleEngine
LIB = os.environ.get("LIB") or None      # LIB=tests/wave_emu/libmpcq_emu.so: the lane emulator (CPU)

def swarm(B, N, nb, K, start, precision=1, seed=1):
    kw = dict(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, dt_pred=0.01)
    if nb: kw.update(basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
    e, o = Engine(EngineConfig(precision=precision, **kw), lib_path=LIB), OracleEngine(EngineConfig(**kw))
    traj, lens = swarm_trajectories(seed, 0, B)
    x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    if start:
        traj, lens = np.ascontiguousarray(traj[:, start:]), lens - start
        x = traj[:, 0].copy()
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    rows = []
    for k in range(K):
        e.set_state(**o.get_state())
        t0 = time.time()
        w, xp = e.step(x); dt = time.time() - t0
        wo, xpo = o.step(x)
        st, it = e.get_status(), e.get_qp_iter()
        err = np.abs(w - wo).max(axis=1) / np.maximum(np.abs(wo).max(axis=1), 1e-2)
        for b in range(B): rows.append((k, b, int(st[b]), int(it[b]), float(err[b])))
        for _ in range(2): x = o.plant_update(x, wo, 5e-3)
    return rows, dt

if __name__ == "__main__":
    B, N, nb, K, start = (int(v) for v in sys.argv[1:6])
    rows, dt = swarm(B, N, nb, K, start)
    r = np.array(rows)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "f32_probe_rows.npy"), r)
    print("last step time", dt)
    for name, sel in (("status0", r[:, 2] == 0), ("flagged", r[:, 2] == 8), ("other", (r[:, 2] != 0) & (r[:, 2] != 8))):
        if sel.any():
            e = r[sel, 4]
            print(name, int(sel.sum()), "worst %.2e p99 %.2e median %.2e" % (e.max(), np.quantile(e, 0.99), np.median(e)))
    bad = r[r[:, 4] > 1e-5]
    for row in bad[np.argsort(-bad[:, 4])][:15]: print("  k %d b %d status %d it %d err %.2e" % tuple(row[:4].astype(int).tolist() + [row[4]]))
