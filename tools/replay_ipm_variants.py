#!/usr/bin/env python3
"""Diagnostic (CPU): the fallback solves collected by tools/dump_multipass.py (gpurun_out/multipass_cases.npz) through the lane emulator under
the tuning environment of the caller (MPCQ_TUNING=1 MPCQ_IPM_RD=.. etc.): factorisations per solve (interior-point iterations + passes) and the
deviation of the control from the GPU's.   usage: replay_ipm_variants.py first last"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace

lib = os.path.join(ROOT, "tests", "wave_emu", "libmpcq_emu.so")
d = np.load(os.path.join(ROOT, "gpurun_out", "multipass_cases.npz"))
lo, hi = int(sys.argv[1]), int(sys.argv[2])
tot, dev, fac, swp = [], [], [], []
for c in range(lo, hi):
    b = int(d["b"][c])
    traj, lens = bench.workload(2026, b, 1, 1000)
    e = Engine(EngineConfig(batch=1, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)), lib_path=lib)
    e.set_trajectories(traj, lens)
    e.set_state(X=d["X"][c][None], U=d["U"][c][None], mu=d["mu"][c][None], C=d["C"][c][None], x_pred_prev=d["xpp"][c][None], has_prev=d["hp"][c:c + 1], idx=d["idx"][c:c + 1])
    e.set_solver_state(qp_iter=d["qp_iter"][c:c + 1])
    w, _ = e.step(d["x"][c][None])
    it = int(e.get_qp_iter()[0]); f, s = e.get_qp_work()
    tot.append(it % 1000); dev.append(float(np.abs(w[0] - d["w"][c]).max())); fac.append(int(f[0])); swp.append(int(s[0]))
    e.close()
print(f"cases {lo}..{hi - 1}: factorisations {tot} mean {np.mean(tot):.2f}; sweeps mean {np.mean(swp):.1f}; chain cost (fac x 632 + swp x 73 ns) x 20 mean {np.mean([(a * 632.2 + b * 73.0) * 20e-3 for a, b in zip(fac, swp)]):.1f} us; max |w - w_gpu| {max(dev):.2e}")
