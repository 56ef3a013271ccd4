import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
B, N, nb = 4, 20, 10
traj, lens = swarm_trajectories(11, 0, B)
x0 = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1))
os.environ["MPCQ_STAGE_MEM"] = "global"
res = {}
for generic in (False, True):
    if generic: os.environ["MPCQ_GENERIC"] = "1"
    else: os.environ.pop("MPCQ_GENERIC", None)
    e = Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb)), lib_path=os.path.join(ROOT, "mpc_quad_ros_amd", sys.argv[1]))
    e.set_trajectories(traj, lens); e.sim_reset(x0)
    e.sim_steps(1, 2, 5e-3)
    n = ctypes.c_int32()
    e.lib.mpcq_debug_stage.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    e.lib.mpcq_debug_stage(e.h, None, ctypes.byref(n))
    out = np.zeros((B, n.value))
    e.lib.mpcq_debug_stage(e.h, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    st = e.get_state()
    res[generic] = (out, st["X"].reshape(B, -1), st["U"].reshape(B, -1), e.get_status())
a, b = res[False], res[True]
nAB = N * 13 * 16 + 16
print("status special", a[3], "generic", b[3])
d = np.abs(a[0] - b[0])
print("stage record n", a[0].shape, "max diff AB", d[:, :nAB].max(), "c", d[:, nAB:nAB + N * 16].max(), "qv", d[:, nAB + N * 16:].max(), "nan generic", np.isnan(b[0]).sum())
print("X diff", np.nanmax(np.abs(a[1] - b[1])), "U diff", np.nanmax(np.abs(a[2] - b[2])), "nan X", np.isnan(b[1]).sum(), "nan U", np.isnan(b[2]).sum())
ix = np.argwhere(np.isnan(b[1])); print("nan X idx", ix[:10].tolist())
dU = np.abs(a[2] - b[2])[0].reshape(N, 4); print("U diff per stage quad0", dU.max(axis=1))
dX = np.abs(a[1] - b[1])[0].reshape(N + 1, 13); print("X diff per stage quad0", np.nanmax(dX, axis=1))
