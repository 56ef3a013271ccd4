import sys, numpy as np
sys.path.insert(0, '.')
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
def run(B, N, nb, K, tune, chunks):
    traj, lens = swarm_trajectories(7, 0, B)
    x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
    res = {}
    for mode in ("sim_steps", "sim_run"):
        e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), tune=tune))
        e.set_trajectories(traj, lens); e.sim_reset(x0)
        out = []
        for c in range(chunks):
            getattr(e, mode)(K // chunks, 2, 5e-3)
            x, w = e.sim_get_state(); st = e.get_state()
            out.append((x.copy(), w.copy(), st["X"].copy(), st["mu"].copy(), e.get_qp_iter().copy()))
        res[mode] = out; e.close()
    for c in range(chunks):
        a, b = res["sim_steps"][c], res["sim_run"][c]
        d = [float(np.abs(a[k] - b[k]).max()) for k in range(4)]
        bad = np.flatnonzero(np.abs(a[1] - b[1]).max(axis=1) > 0)
        print(f"  chunk {c}: max|dx| {d[0]:.3e} |dw| {d[1]:.3e} |dX| {d[2]:.3e} |dmu| {d[3]:.3e}  quads differing in w: {len(bad)} {bad[:8].tolist()}  qp_iter {a[4][bad[:4]].tolist()} vs {b[4][bad[:4]].tolist()}", flush=True)
import os
CASES = ((1024, 20, 20, None), (1024, 50, 50, None), (1024, 20, 10, None)) if len(sys.argv) < 2 else ((1024, 20, 20, None),)
print('library', os.environ.get('MPCQ_LIB', 'libmpcq.so'), flush=True)
for (B, N, nb, tune) in CASES:
    print(B, N, nb, tune, flush=True)
    run(B, N, nb, 16, tune, 4)
