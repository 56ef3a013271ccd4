#!/usr/bin/env python3
"""BASELINE configs[4] ablation (GPU box): long horizon N=50, 50 RGP basis points per axis, batch 4096 --
fp64 / fp32 arithmetic, bf16 STORAGE of the records (fp32 arithmetic) and the matrix cores switched off, each as
(a) worst relative control deviation against the fp64 CPU oracle over a host-driven closed loop of 64 quadrotors x 60
control periods and (b) lockstep control steps/s at B = 4096.  Needs the ablation builds:
  make -C mpc_quad_ros_amd/csrc variant NAME=nomfma SHAPES=50_50 EXTRA="-DMPCQ_NO_MFMA '-DMPCQ_SHAPE_LIST(X)=X(50,50)'"
  make -C mpc_quad_ros_amd/csrc variant NAME=bf16 SHAPES=50_50 EXTRA="-DMPCQ_BF16_RECORDS '-DMPCQ_SHAPE_LIST(X)=X(50,50)'"
The oracle is used here as the checker only (tools/ is not the product)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
from oracle.oracle import OracleEngine

N, NB = int(os.environ.get("ABL_N", 50)), int(os.environ.get("ABL_NB", 50))
BP, KP = int(os.environ.get("ABL_BP", 64)), int(os.environ.get("ABL_KP", 60))
BT, KT = int(os.environ.get("ABL_BT", 4096)), int(os.environ.get("ABL_KT", 40))
X0 = np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
lib = lambda n: None if n == "product" else os.path.join(ROOT, "mpc_quad_ros_amd", f"libmpcq_{n}.so")
kw = dict(N=N, quad=hummingbird(), nb=NB, basis=rgp_basis_linspace(12.0, NB))


def parity(name, precision):
    e = Engine(EngineConfig(batch=BP, precision=precision, **kw), lib_path=lib(name))
    o = OracleEngine(EngineConfig(batch=BP, **kw))
    traj, lens = swarm_trajectories(1, 0, BP)
    e.set_trajectories(traj, lens); o.set_trajectories(traj, lens)
    x = np.tile(X0, (BP, 1))
    worst, typ, failed, changes = 0.0, [], 0, 0
    for k in range(KP):
        w, _ = e.step(x); wo, _ = o.step(x)
        ok = (e.get_status() & 7) == 0          # (MPCQ_SOLVE_LOW_ACCURACY, 8, is a warning of the fp32 mode: counted as solved here)
        failed += int((~ok).sum())
        changes += int(((e.get_qp_iter() % 1000) > 1).sum())
        err = np.abs(w - wo).max(axis=1) / np.maximum(np.abs(wo).max(axis=1), 1e-2)      # per quadrotor
        worst = max(worst, float(err[ok].max()) if ok.any() else 0.0); typ.append(float(np.median(err)))
        x = o.plant_control_period(x, wo, 0.01, 5e-3)[0]
    return dict(worst_rel_dev=worst, median_rel_dev=float(np.median(typ)), failed_solves=failed, quad_steps_with_working_set_change=changes)


def throughput(name, precision):
    e = Engine(EngineConfig(batch=BT, precision=precision, **kw), lib_path=lib(name))
    traj, lens = swarm_trajectories(2026, 0, BT)
    e.set_trajectories(traj, lens); e.sim_reset(np.tile(X0, (BT, 1)))
    e.sim_steps(60, 2, 5e-3)
    t0 = time.perf_counter(); e.sim_steps(KT, 2, 5e-3); e.synchronize(); t1 = time.perf_counter()
    kt, kl = e.get_kernel_time()
    return dict(steps_per_s=BT * KT / (t1 - t0), ms_per_step=1e3 * (t1 - t0) / KT, kernel_avg_ms=1e3 * kt / max(kl, 1), failed=int(((e.get_status() & 7) != 0).sum()))


rows = []
for label, name, precision in [("fp64", "product", 0), ("fp32", "product", 1), ("fp32 arithmetic, bf16 storage of records + RGP state", "bf16", 1),
                               ("fp64, matrix cores off", "nomfma", 0), ("fp32, matrix cores off", "nomfma", 1),
]:
    if name != "product" and not os.path.exists(lib(name)):
        print("skip", label, "(build missing)"); continue
    r = dict(config=label, N=N, nb=NB)
    r["parity"] = parity(name, precision) if BP else None
    r["throughput"] = throughput(name, precision) if BT else None
    rows.append(r)
    print(json.dumps(r), flush=True)
with open(os.path.join(ROOT, "gpurun_out", os.environ.get("ABL_OUT", "ablation_config5.json")), "w") as f:
    json.dump(dict(shape=dict(N=N, nb=NB, batch_parity=BP, periods_parity=KP, batch_throughput=BT, periods_throughput=KT), rows=rows), f, indent=1)
