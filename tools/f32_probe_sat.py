"""Diagnostic: the saturating-references case of tests/parity_cases.py in f32, per quadrotor and step.  usage: f32_probe_sat.py B K"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine
LIB = os.environ.get("LIB") or None
make = lambda cfg: Engine(cfg, lib_path=LIB)
B, K = int(sys.argv[1]), int(sys.argv[2])
# teacher-forced variant of case_saturating_references with per-step error
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from oracle.oracle import OracleEngine
N, nb = 20, 10
kw = dict(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb))
e, o = make(EngineConfig(precision=1, **kw)), OracleEngine(EngineConfig(**kw))
T = 60 + K * 5
traj = np.zeros((B, T, 13)); traj[:, :, 3] = 1.0
t = np.arange(T) * 0.01
for b in range(B):
    A, w = 2.0 + b, 3.0 + 0.7 * b
    traj[b, :, 0] = A * np.sin(w * t); traj[b, :, 7] = A * w * np.cos(w * t)
    traj[b, :, 2] = 3.0 + 1.5 * np.sign(np.sin(2.0 * t + b))
lens = np.full(B, T, dtype=np.int32)
e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
x = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
rows = []
for k in range(K):
    w_e, _ = e.step(x); w_o, _ = o.step(x)
    st, it = e.get_status(), e.get_qp_iter()
    err = np.abs(w_e - w_o).max(axis=1)
    for b in range(B): rows.append((k, b, int(st[b]), int(it[b]), float(err[b])))
    x = o.plant_control_period(x, w_o, 0.01, 5e-3)[0]
    s = o.get_state()
    e.set_state(X=s["X"], U=s["U"], mu=s["mu"], C=s["C"], x_pred_prev=s["x_pred_prev"], has_prev=s["has_prev"], idx=s["idx"])
r = np.array(rows)
print("statuses", {int(v): int((r[:, 2] == v).sum()) for v in np.unique(r[:, 2])})
ok = r[:, 2] == 0
print("status0 worst %.2e p99 %.2e median %.2e" % (r[ok, 4].max(), np.quantile(r[ok, 4], 0.99), np.median(r[ok, 4])))
bad = r[r[:, 4] > 1e-5]
for row in bad[np.argsort(-bad[:, 4])][:15]: print("  k %d b %d status %d it %d err %.2e" % tuple(row[:4].astype(int).tolist() + [row[4]]))
