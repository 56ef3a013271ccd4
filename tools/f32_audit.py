#!/usr/bin/env python3
"""Diagnostic (GPU box): EVERY solve of an f32 (mixed-precision) lockstep run of the bench workload against the fp64 engine on the same inputs
(tests/parity_cases.py: case_f32_every_solve_against_f64 -- the GPU test test_f32_every_solve_against_the_fp64_engine runs a short one).
What it answers: does any solve with status 0 miss the 1e-4 budget (a silent miss), and what do the flagged ones (MPCQ_SOLVE_LOW_ACCURACY) look like.
usage: [SOAK_B= SOAK_N= SOAK_NB=] f32_audit.py periods seed [preroll]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine
K, seed = int(sys.argv[1]), int(sys.argv[2])
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B, N, nb = int(os.environ.get('SOAK_B', 1024)), int(os.environ.get('SOAK_N', 20)), int(os.environ.get('SOAK_NB', 10))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
r = pc.case_f32_every_solve_against_f64(lambda cfg: Engine(cfg), B, N, nb, K, seed, pre, dump_prefix=os.path.join(ROOT, "gpurun_out", f"f32_audit_hit_{seed}_{N}"))
print(f"f32 vs f64, every solve: B = {B}, N = {N}, nb = {nb}, seed {seed}, {K} periods behind a pre-roll of {pre}")
for key, name in (("clean", "status 0"), ("flagged", "status 8 (MPCQ_SOLVE_LOW_ACCURACY)"), ("failed", "failed")):
    print(f"  {name}: {r[key]['solves']} solves, worst deviation {r[key]['worst']:.3e}, beyond 1e-4: {r[key]['beyond']}")
print(f"  status-0 fallback solves: worst deviation {r['fallback_clean_worst']:.3e}")
print("  flagged or beyond 1e-4 (period, quadrotor, status, qp_iter, deviation relative to own largest control, absolute deviation, own largest control):", r["hits"][:60])
