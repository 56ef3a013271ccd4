"""Diagnostic: the free_running_equals_lockstep case of tests/test_engine_edges.py, per precision / kernel flavour, with the
quadrotors and arrays that differ."""
import sys, numpy as np
sys.path.insert(0, '.')
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
B, N, nb, K = 64, 20, 10, 40
traj, lens = swarm_trajectories(5, 0, B)
lens = lens.copy(); lens[0] = 6
x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
for precision in (0, 1):
    for tune in (None, dict(generic_kernel=1)):
        res = []
        for mode in ("sim_steps", "sim_run"):
            e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=precision, tune=tune))
            e.set_trajectories(traj, lens); e.sim_reset(x0)
            hist = []
            for k in range(K):
                getattr(e, mode)(1, 2, 5e-3)
                hist.append(e.sim_get_state()[0].copy())
            res.append(np.array(hist)); e.close()
        d = np.abs(res[0] - res[1]).max(axis=2)      # [K, B]
        first = np.argmax(d.max(axis=1) > 0) if (d > 0).any() else -1
        print(f"precision {precision} tune {tune}: max diff {d.max():.3e}, first differing period {first}, quads differing at the end {np.flatnonzero(d[-1] > 0)[:10].tolist()} of {int((d[-1] > 0).sum())}", flush=True)
        if first >= 0:
            q = int(np.argmax(d[first]))
            print("   quad", q, "period", first, "lockstep x", res[0][first, q, :3], "run x", res[1][first, q, :3], flush=True)
