#!/usr/bin/env python3
"""Diagnostic (CPU): replays the quadrotor-steps collected by tools/dump_multipass.py through the lane emulator built with
the pass-by-pass log (make -C tests/wave_emu debug) and prints what each pass of the active-set method pinned / released."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories

subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "wave_emu"), "debug"], stdout=subprocess.DEVNULL)
lib = os.path.join(ROOT, "tests", "wave_emu", "libmpcq_emu_dbg.so")
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "multipass_cases.npz"))
which = [int(a) for a in sys.argv[2:]] or range(len(d["b"]))
N, nb = 20, 10
for c in which:
    b = int(d["b"][c])
    import bench
    traj, lens = bench.workload(2026, b, 1, 1000)
    e = Engine(EngineConfig(batch=1, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb)), lib_path=lib)
    e.set_trajectories(traj, lens)
    e.set_state(X=d["X"][c][None], U=d["U"][c][None], mu=d["mu"][c][None], C=d["C"][c][None], x_pred_prev=d["xpp"][c][None],
                has_prev=d["hp"][c:c + 1], idx=d["idx"][c:c + 1])
    e.set_solver_state(qp_iter=d["qp_iter"][c:c + 1])
    print(f"=== case {c}: quadrotor {b}, control period {int(d['step'][c])}, {int(d['passes'][c])} factorisations on the GPU", flush=True)
    U = d["U"][c]
    sat = [(i, j, U[i, j]) for i in range(N) for j in range(4) if U[i, j] <= 0.0 or U[i, j] >= 1.0]
    print("    iterate U at a bound:", " ".join(f"s{i}r{j}={'0' if v <= 0 else '1'}" for i, j, v in sat), flush=True)
    w, _ = e.step(d["x"][c][None])
    print(f"    emulator passes {int(e.get_qp_iter()[0])}, |w - w_gpu| {np.abs(w[0] - d['w'][c]).max():.2e}", flush=True)
    e.close()
