#!/usr/bin/env python3
"""Diagnostic (GPU box): do the any-shape fp64 step kernels built at -O3 agree with the shape-specialised ones?
Drives a libmpcq build through raw ctypes (works with the round-1 ABI too), 64 quadrotors x 210 closed-loop periods,
four variants (stage records in LDS / global memory x specialised / any-shape instance).
usage: o3_discrepancy_probe.py path/to/libmpcq_*.so"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories

os.environ["MPCQ_TUNING"] = "1"     # the library reads MPCQ_STAGE_MEM / MPCQ_GENERIC only under MPCQ_TUNING=1 (since 0.3): without it the four variants below would be one
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int32)
lib.mpcq_create_sized.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]   # (the exported mpcq_create reads the 0.3 layout only)
lib.mpcq_set_trajectories.argtypes = [ctypes.c_void_p, dp, ip, ctypes.c_int32]
lib.mpcq_sim_reset.argtypes = [ctypes.c_void_p, dp]
lib.mpcq_sim_steps.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double]
lib.mpcq_sim_get_state.argtypes = [ctypes.c_void_p, dp, dp]
lib.mpcq_get_status.argtypes = [ctypes.c_void_p, ip]
lib.mpcq_destroy.argtypes = [ctypes.c_void_p]
B, N, nb, K = 64, 20, 10, 210
traj, lens = swarm_trajectories(11, 0, B)
traj = np.ascontiguousarray(traj); lens = np.ascontiguousarray(lens, dtype=np.int32)
x0 = np.ascontiguousarray(np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]), (B, 1)))
out = {}
for mem in ("lds", "global"):
    for generic in (False, True):
        os.environ["MPCQ_STAGE_MEM"] = mem
        if generic:
            os.environ["MPCQ_GENERIC"] = "1"
        else:
            os.environ.pop("MPCQ_GENERIC", None)
        cfg = EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb))
        c = cfg.to_c()
        h = ctypes.c_void_p()
        assert lib.mpcq_create_sized(ctypes.byref(c), ctypes.sizeof(c), ctypes.byref(h)) == 0
        lib.mpcq_set_trajectories(h, traj.ctypes.data_as(dp), lens.ctypes.data_as(ip), traj.shape[1])
        lib.mpcq_sim_reset(h, x0.ctypes.data_as(dp))
        ws, bad = [], 0
        for k in range(K):
            lib.mpcq_sim_steps(h, 1, 2, 5e-3)
            w, x, st = np.zeros((B, 4)), np.zeros((B, 13)), np.zeros(B, np.int32)
            lib.mpcq_sim_get_state(h, x.ctypes.data_as(dp), w.ctypes.data_as(dp))
            lib.mpcq_get_status(h, st.ctypes.data_as(ip))
            bad += int((st != 0).sum())
            ws.append(w)
        out[(mem, generic)] = (np.array(ws), bad)
        lib.mpcq_destroy(h)
ref = out[("global", False)][0]
for k, (v, bad) in out.items():
    print(f"{os.path.basename(sys.argv[1])}: stage records {k[0]:6s} {'any-shape  ' if k[1] else 'specialised'}  max |w - w(global, specialised)| {np.nanmax(np.abs(v - ref)):.3e}  "
          f"NaN controls {int(np.isnan(v).sum())}  failed solves {bad}")
