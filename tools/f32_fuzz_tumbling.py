#!/usr/bin/env python3
"""Diagnostic (GPU box): the f32 (mixed-precision) mode at the edge of its validity, fuzzed.  Periods 95 .. 129 of the reference's tumbling
traj2_v10_a10_gp2 flight: the oracle runs the flight on the logged measurements; before every period B copies of its state go into an f32 and an
fp64 engine, each copy gets the logged measurement plus its own perturbation (sigma per component, copy 0 unperturbed), both engines solve, the
controls are compared per copy.  What it answers: how often does a solve with status 0 miss 1e-4 when the inputs move in the last digits.
usage: f32_fuzz_tumbling.py [B] [sigma]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import config_for_log, load_golden
from oracle.oracle import OracleEngine
from mpc_quad_ros_amd.engine import Engine
import dataclasses
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
g = load_golden("log_traj2_v10_a10_gp2_whole.npz")
cfg1 = config_for_log(g)
o = OracleEngine(cfg1); o.set_trajectories(g["x_ref"][None])
traj = np.repeat(g["x_ref"][None], B, axis=0)
e32 = Engine(dataclasses.replace(config_for_log(g, precision=1), batch=B)); e64 = Engine(dataclasses.replace(config_for_log(g), batch=B))
for e in (e32, e64):
    e.set_trajectories(traj)
rng = np.random.default_rng(0)
tot = {"clean": 0, "flagged": 0, "failed": 0, "miss": 0}
worst_clean = 0.0
for k in range(130):
    if k >= 95:
        st = {name: np.repeat(v, B, axis=0) for name, v in o.get_state().items()}
        x = np.repeat(g["x_odom"][k][None], B, axis=0) + rng.normal(0, sigma, (B, 13)) * (np.arange(B)[:, None] > 0)
        e32.set_state(**st); e64.set_state(**st)
        w32, _ = e32.step(x); w64, _ = e64.step(x)
        s32, s64 = e32.get_status(), e64.get_status()
        ok64 = s64 == 0
        dev = np.abs(w32 - w64).max(axis=1) / max(np.abs(w64).max(), 1e-3)
        clean, flagged, failed = (s32 == 0) & ok64, (s32 == 8) & ok64, ((s32 & 7) != 0) & ok64
        miss = clean & (dev > 1e-4)
        tot["clean"] += int(clean.sum()); tot["flagged"] += int(flagged.sum()); tot["failed"] += int(failed.sum()); tot["miss"] += int(miss.sum())
        if clean.any():
            worst_clean = max(worst_clean, float(dev[clean].max()))
        print(f"period {k}: fp64 ok {int(ok64.sum())}/{B}; f32 status 0: {int(clean.sum())} (worst {dev[clean].max() if clean.any() else 0:.2e}, beyond 1e-4: {int(miss.sum())}), flagged {int(flagged.sum())}, failed {int(failed.sum())}")
    o.step(g["x_odom"][k][None])
print(f"total over periods 95 .. 129, {B} copies, sigma {sigma:g}: {tot}, worst status-0 deviation {worst_clean:.2e}")
