"""Diagnostic (GPU box): the f32 parity_on_workload leg of a bench configuration, per quadrotor and period, worst offenders first; the
state in front of the worst solve is saved for a single-quadrotor replay.  usage: f32_workload_probe.py B N nb preroll quads periods"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpc_quad_ros_amd.engine import Engine, qp_fallback
from mpc_quad_ros_amd.params import PRECISION_F32, EngineConfig, hummingbird, rgp_basis_linspace
from oracle.oracle import OracleEngine
B, N, nb, pre, Q, K = (int(v) for v in sys.argv[1:7])
refs = bench.workload(2026, 0, B, pre + 5 + 20 + K + 150)
e, _ = bench.make_engine(B, N, nb, PRECISION_F32, 0, 0, 0, refs=refs)
n_sub = e.plant_substeps(0.01, 5e-3)
e.sim_run(pre, n_sub, 5e-3); e.sim_steps(25, n_sub, 5e-3)
dump = bench.dump_engine(e); e.close()
it0 = dump["solver"]["qp_iter"]
fb = np.flatnonzero(qp_fallback(it0))
sel = np.sort(np.concatenate([fb, np.setdiff1d(np.arange(len(it0)), fb)])[:Q])
kw = dict(batch=len(sel), N=N, T=1.0, quad=hummingbird(), nb=nb, dt_pred=0.01, basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
o = OracleEngine(EngineConfig(**kw)); e = Engine(EngineConfig(precision=PRECISION_F32, **kw))
traj, lens = np.ascontiguousarray(refs[0][sel]), np.ascontiguousarray(refs[1][sel])
st = {k: np.ascontiguousarray(v[sel]) for k, v in dump["state"].items()}
sol = {k: np.ascontiguousarray(v[sel]) for k, v in dump["solver"].items()}
o.set_trajectories(traj, lens); o.set_state(**st)
e.set_trajectories(traj, lens); e.set_state(**st); e.set_solver_state(**sol)
x = dump["x"][sel].copy()
rows, saved = [], None
for k in range(K):
    so = o.get_state()
    e.set_state(**so)
    prev = e.get_qp_iter().copy()
    w, _ = e.step(x); wo, _ = o.step(x)
    err = np.abs(w - wo).max(axis=1); rel = err / np.maximum(np.abs(wo).max(axis=1), 1e-2)
    it, stt = e.get_qp_iter(), e.get_status()
    for b in range(len(sel)):
        rows.append((k, b, int(stt[b]), int(it[b]), float(err[b]), float(rel[b]), float(np.abs(wo[b]).max())))
    b = int(np.argmax(rel))
    if saved is None or rel[b] > saved[0]:
        saved = (float(rel[b]), dict(k=k, b=b, x=x[b].copy(), traj=traj[b], len=lens[b], prev=prev[b], w=w[b], wo=wo[b], **{f"st_{n}": v[b] for n, v in so.items()}))
    x = o.plant_control_period(x, wo, 0.01, 5e-3)[0]
r = np.array(rows)
print("worst abs %.2e worst per-quad rel %.2e" % (r[:, 4].max(), r[:, 5].max()))
for row in r[np.argsort(-r[:, 5])][:10]:
    print("  k %d b %d status %d it %d abs %.2e rel %.2e |w|max %.3f" % (row[0], row[1], row[2], row[3], row[4], row[5], row[6]))
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/f32_workload_worst.npz", N=N, nb=nb, **saved[1])
