import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, config_for_log
from mpc_quad_ros_amd.engine import Engine
from oracle.oracle import OracleEngine
name = sys.argv[1] if len(sys.argv) > 1 else "log_traj1_v10_a10_gp0.npz"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = load_golden(name)
cfg = config_for_log(g, precision=prec)
e, o = Engine(cfg, lib_path=os.environ.get("DBG_LIB")), OracleEngine(config_for_log(g))
e.set_trajectories(g["x_ref"][None]); o.set_trajectories(g["x_ref"][None])
np.set_printoptions(linewidth=200, precision=2)
for k in range(K):
    e.set_state(**o.get_state())
    w, xp = e.step(g["x_odom"][k][None]); wo, xpo = o.step(g["x_odom"][k][None])
    se, so = e.get_state(), o.get_state()
    dX = np.abs(se["X"][0] - so["X"][0]).max(axis=1); dU = np.abs(se["U"][0] - so["U"][0]).max(axis=1)
    print(k, "w", np.abs(w - wo).max(), "cost", e.get_cost()[0], o.get_cost()[0], "it", e.get_qp_iter()[0], o.get_qp_iter()[0], "st", e.get_status()[0])
    print("   dX per node", dX)
    print("   dU per node", dU)
    if k == 0:
        print("X engine node1", se["X"][0][1]); print("X oracle node1", so["X"][0][1])
        print("X engine node5", se["X"][0][5]); print("X oracle node5", so["X"][0][5])
