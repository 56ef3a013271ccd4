#!/usr/bin/env python3
"""bench.py — batched MPC+RGP control steps/s (BASELINE.json metric) on MI355X.

One "step" = one fused control step (reference chunk -> SQP-RTI solve -> first input -> nominal
prediction -> drag estimate -> 3 RGP updates) for every quadrotor of the batch, followed by the
on-device drag plant that produces the next measurement (closed loop, no host traffic: all inputs
are resident in HBM when the timed region starts).  Workload at N GPUs: BASELINE configs[1]
per GPU (1024 hummingbirds, horizon N=20, 10 RGP basis points per axis) on seeded random-waypoint
min-snap trajectories (mpc_quad_ros_amd/csrc/minsnap.cpp); instances are sharded by global index, no collective on the data path, one RCCL
all-reduce of the 5-number tracking statistic at the end (weak scaling).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpc_quad_ros_amd.engine import (Engine, WARM_PINS, WARM_WRONG, qp_fallback, qp_flip, qp_passes,  # noqa: E402
                                      qp_warm_exit)
from mpc_quad_ros_amd.params import PRECISION_F32, PRECISION_F64, EngineConfig, hummingbird, rgp_basis_linspace  # noqa: E402
from mpc_quad_ros_amd.trajectories import swarm_missions  # noqa: E402

X0 = np.array([0, 0, 3.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP_VECTOR_PEAK = {"f64": 78.6, "f32": 157.3}   # TFLOP/s (matrix = vector rate for these dtypes on gfx950)


def algorithmic_bytes(N, nb, itemsize):
    """floats(N,nb) = 13N + 2(17N+13) + 6nb + 6nb^2 + 45 per quad per step."""
    return itemsize * (13 * N + 2 * (17 * N + 13) + 6 * nb + 6 * nb * nb + 45)


def algorithmic_flops(N, nb, passes):
    """Useful flops of the implemented algorithm per quad per step: shooting
    N*(4*2*60*14 + 4*(150+30*nb)); per QP pass one Riccati factorisation N*2*(13*13*14 + 14*15/2*13 +
    13*13*4 + 60) and four vector sweeps 4*N*2*(13*14+4*13); RGP 3*8*nb^2.  (The matrix-core tiles
    execute 9*2048 + 16*2048 padded flops per stage and pass; only the useful ones are counted.)"""
    shoot = N * (4 * 2 * 60 * 14 + 4 * (150 + 30 * nb))
    fact = N * 2 * (13 * 13 * 14 + 105 * 13 + 13 * 13 * 4 + 60)
    vec = 4 * N * 2 * (13 * 14 + 4 * 13)
    return shoot + passes * (fact + vec) + 24 * nb * nb


PREROLL = 600                 # un-timed control periods before the warm-up (see --preroll)
NONQP_CHAIN_US = 33.7         # load + shooting + full step + post phase of one quadrotor at their MEASURED cost (80.9 k cycles at 2.4 GHz,
                              # profiles/r3_phase_cycles.txt): not floored, so latency_roofline.frac errs on the high side


def latency_roofline(e, n_sub, N, launches=20):
    """What bounds a lockstep launch is not bytes but the dependent instruction chain of its slowest quadrotor.  Floor of that
    chain: (factorisations x N stages x the measured latency of one factorisation stage's dependent chain) + (sweeps x N x the
    measured chain of one sweep stage) -- both from tools/microbench/chain_floor.hip (registers only, nothing but the chain;
    profiles/r5_chain_floor.json) -- + the non-QP phases at their measured cost.  The interior-point iterations of a fallback solve
    of the fp64 instances run in float (csrc/mpcq_kernels.hpp ipm_float_stage): their factorisation and their three sweeps are priced
    with the float chains -- exactly the iterations the engine reports as float in its work word (mpcq_get_qp_work bits 27..31; advisor finding of
    round 5: the pricing used to be inferred from the shape).  `launches`
    further lockstep periods, one call each, with the work counters (mpcq_get_qp_work) and the launch time read back after every one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r5_chain_floor.json")) as fh:
            fl = json.load(fh)
    except (OSError, ValueError):
        return None
    floor, got, slow_fac, slow_ipm, float_any = [], [], [], [], False
    for _ in range(launches):
        e.sim_steps(1, n_sub, 5e-3)
        kt, _kl = e.get_kernel_time()
        fac, swp = e.get_qp_work()
        # interior-point iterations of a solve: a fallback solve executes 2 it + 2 (+1) more sweeps than factorisations (solve_qp's work count)
        ipm = np.where(qp_fallback(e.get_qp_iter()), np.maximum(0, (swp - fac - 2) // 2), 0)
        ipm_f = np.minimum(e.get_qp_float_iterations(), ipm)      # those the engine says it ran in float (the work word, bits 27..31): priced with the float chains
        float_any = float_any or bool(ipm_f.any())
        chain_ns = ((fac - ipm_f) * fl["factor_stage_chain_ns"] + ipm_f * fl["factor_stage_chain_f32_ns"]
                    + (swp - 3 * ipm_f) * fl["sweep_stage_chain_ns"] + 3 * ipm_f * fl["sweep_stage_chain_f32_ns"]) * N
        floor.append(float(chain_ns.max()) * 1e-6 + NONQP_CHAIN_US * 1e-3)
        got.append(1e3 * kt)
        b = int(np.argmax(chain_ns))
        slow_fac.append(int(fac[b]))
        slow_ipm.append(int(ipm[b]))
    return {"bound": "dependent-chain latency of the slowest quadrotor of a launch", "floor_ms": float(np.mean(floor)), "achieved_ms": float(np.mean(got)),
            "frac": float(np.mean(floor) / np.mean(got)), "launches": launches, "slowest_quad_factorisations_mean": float(np.mean(slow_fac)),
            "slowest_quad_interior_point_iterations_mean": float(np.mean(slow_ipm)), "interior_point_iterations_in_float": bool(float_any),
            "factor_stage_chain_ns": fl["factor_stage_chain_ns"], "sweep_stage_chain_ns": fl["sweep_stage_chain_ns"],
            "factor_stage_chain_f32_ns": fl["factor_stage_chain_f32_ns"], "sweep_stage_chain_f32_ns": fl["sweep_stage_chain_f32_ns"],
            "non_qp_phases_us_measured": NONQP_CHAIN_US,
            "note": "floor = chain latencies measured in isolation (registers only, one wavefront per SIMD); the product's stage additionally loads its "
                    "operands, hands rows over through LDS, stores gains / cost-to-go and carries ~4x the instructions of the bare chain"}


def workload(seed, first_index, B, periods):
    """Continuous operation of the node, per quadrotor: min-snap flights through 3 random waypoints (v_max = a_max = 12, the
    launch defaults), each requested from the end point of the previous one (mpc_quad_ros_amd.trajectories.minsnap_mission),
    long enough for `periods` control periods plus one horizon."""
    return swarm_missions(seed, first_index, B, periods + 150, max_rows=periods + 150, v_max=12.0, a_max=12.0)


def kernel_source_sha16():
    """Identity of the device code a profile was taken on: hash of the kernel / host sources and the build recipe (the GPU
    box has no .git).  PMC traffic files carry it; a file from another build is not attached to this run's line."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mpcq_kernels.hpp", "mpcq_api.hip", "mpcq_spec.hip", "Makefile", "cc_checked.sh"):
        with open(os.path.join(ROOT, "mpc_quad_ros_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def dump_engine(e):
    """Everything a continued run needs: iterate / RGP state / cursors, warm-start flags and statistics, plant states."""
    return {"state": e.get_state(), "solver": e.get_solver_state(), "x": e.sim_get_state()[0]}


def parity_on_workload(dump, refs, N, nb, device, quads=64, periods=30, free_running=False, precision=PRECISION_F64, lib_path=None):
    """Oracle parity ON THE BENCH WORKLOAD (the checker, never the thing measured): a sample of `quads` quadrotors of a
    pre-rolled engine (`dump` = dump_engine of it, `refs` its references) -- every quadrotor whose last solve went through
    the interior point, filled up from the low indices -- is continued host-driven for `periods` control periods on a small
    engine next to the fp64 CPU oracle: same measurements (the oracle's drag plant), the engine's state overwritten with the
    oracle's before every step (teacher-forced: each step isolates the arithmetic of one solve, warm-start flags stay the
    engine's own).  With free_running a second engine runs alongside WITHOUT being overwritten, and so does a twin of the
    oracle whose iterate was perturbed by 1e-10 (relative) at the start: on this workload some quadrotors (saturated, off
    their reference by decimetres) sit where the RTI iteration itself amplifies any difference -- the twin measures that
    amplification per quadrotor, and the free-running engine is judged against it.  Mirrors the loop body
    src/mpc_controller_node.py:278-318 on both sides.  Returns the worst relative control deviations and what the
    window covered."""
    from oracle.oracle import OracleEngine
    it0 = dump["solver"]["qp_iter"]
    fb = np.flatnonzero(qp_fallback(it0))
    sel = np.sort(np.concatenate([fb, np.setdiff1d(np.arange(len(it0)), fb)])[:quads])
    B = len(sel)
    kw = dict(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, dt_pred=0.01)
    if nb:
        kw.update(basis=rgp_basis_linspace(12.0, nb), theta=[1.0, 0.1, 0.1])
    o = OracleEngine(EngineConfig(**kw))
    engines = [Engine(EngineConfig(device=device, precision=precision, **kw), lib_path=lib_path)]
    twin = None
    if free_running:
        engines.append(Engine(EngineConfig(device=device, precision=precision, **kw), lib_path=lib_path))
        twin = OracleEngine(EngineConfig(**kw))
    traj, lens = np.ascontiguousarray(refs[0][sel]), np.ascontiguousarray(refs[1][sel])
    st = {k: np.ascontiguousarray(v[sel]) for k, v in dump["state"].items()}
    sol = {k: np.ascontiguousarray(v[sel]) for k, v in dump["solver"].items()}
    o.set_trajectories(traj, lens); o.set_state(**st)
    if twin is not None:
        rng = np.random.default_rng(7)
        stp = dict(st)
        stp["X"] = st["X"] * (1.0 + 1e-10 * rng.standard_normal(st["X"].shape))
        stp["U"] = np.clip(st["U"] * (1.0 + 1e-10 * rng.standard_normal(st["U"].shape)), 0.0, 1.0)
        twin.set_trajectories(traj, lens); twin.set_state(**stp)
        dev_free, dev_twin = np.zeros(B), np.zeros(B)
    for e in engines:
        e.set_trajectories(traj, lens); e.set_state(**st); e.set_solver_state(**sol)
    x = dump["x"][sel].copy()
    rel = lambda a, b, floor: float(np.abs(a - b).max() / max(np.abs(b).max(), floor))
    relq = lambda a, b: float((np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-2)).max())
    out = {"max_rel_dev": 0.0, "max_rel_dev_per_quad": 0.0, "rgp_max_rel_dev": 0.0, "quad_steps": 0, "fallbacks": 0, "flip_marked": 0,
           "warm_exit_pins": 0, "warm_exit_wrong": 0, "multi_pass": 0, "failed": 0, "saturated_controls": 0,
           "quads": B, "periods": periods, "fallback_quads_at_dump": int(len(fb)), "mode": "teacher-forced, fp64 CPU oracle"}
    if free_running:
        out["free_running_max_rel_dev"] = 0.0
    e = engines[0]
    for _ in range(periods):
        e.set_state(**o.get_state())
        w, _xp = e.step(x)
        wo, _ = o.step(x)
        its = e.get_qp_iter()
        out["max_rel_dev"] = max(out["max_rel_dev"], rel(w, wo, 1e-3))
        out["max_rel_dev_per_quad"] = max(out["max_rel_dev_per_quad"], relq(w, wo))
        if nb:
            (mu, C), (muo, Co) = e.get_rgp(), o.get_rgp()
            out["rgp_max_rel_dev"] = max(out["rgp_max_rel_dev"], rel(mu, muo, 1.0), rel(C, Co, 1e-2))
        if free_running:
            wf, _ = engines[1].step(x)
            wt, _ = twin.step(x)
            dev_free = np.maximum(dev_free, np.abs(wf - wo).max(axis=1))
            dev_twin = np.maximum(dev_twin, np.abs(wt - wo).max(axis=1))
        out["quad_steps"] += B
        out["fallbacks"] += int(qp_fallback(its).sum())
        out["flip_marked"] += int(qp_flip(its).sum())
        out["warm_exit_pins"] += int((qp_warm_exit(its) == WARM_PINS).sum())
        out["warm_exit_wrong"] += int((qp_warm_exit(its) == WARM_WRONG).sum())
        out["multi_pass"] += int((qp_passes(its) > 1).sum())
        out["failed"] += int(((e.get_status() & 7) != 0).sum())
        out["saturated_controls"] += int(((wo <= 0.0) | (wo >= 1.0)).sum())
        x = o.plant_control_period(x, wo, 0.01, 5e-3)[0]
    if free_running:
        # controls live in [0, 1]: absolute deviations are relative to full thrust.  A quadrotor is "contractive" over the window when
        # the oracle's own 1e-10 twin stays within 1e-7 of it; the others amplify ANY difference (the twin shows by how much).
        calm = dev_twin < 1e-7
        out["free_running_max_rel_dev"] = float(dev_free[calm].max()) if calm.any() else 0.0
        out["free_running_contractive_quads"] = int(calm.sum())
        out["free_running_sensitive_quads"] = {"count": int((~calm).sum()), "engine_max_dev": float(dev_free[~calm].max()) if (~calm).any() else 0.0,
                                               "oracle_twin_max_dev": float(dev_twin[~calm].max()) if (~calm).any() else 0.0,
                                               "engine_over_twin_worst_ratio": float((dev_free[~calm] / np.maximum(dev_twin[~calm], 1e-300)).max()) if (~calm).any() else 0.0}
        twin.close()
    for e in engines:
        e.close()
    o.close()
    return out


def lockstep_leg(e, n_sub, warmup, steps, dist=None):
    """Warm-up, then `steps` lockstep periods timed between two barriers (stream synchronised on both sides, plus the host
    group's barrier under WORLD_SIZE > 1).  Returns (seconds between the barriers, seconds until this rank's own stream was
    idle, mean step-kernel launch time by HIP events)."""
    def barrier():
        e.synchronize()                      # hipStreamSynchronize on the engine's stream (the only one used)
        t = time.perf_counter()
        if dist is not None:
            dist.barrier()
        return t
    e.sim_steps(warmup, n_sub, 5e-3)
    barrier()
    t0 = time.perf_counter()
    e.sim_steps(steps, n_sub, 5e-3)          # K fused steps + plant, back-to-back on one stream
    t_own = barrier()
    t1 = time.perf_counter()
    kt, kl = e.get_kernel_time()             # HIP events around every 4th step-kernel launch (every launch when steps < 8)
    return t1 - t0, t_own - t0, kt / max(kl, 1)


def config_leg(name, refs, B, N, nb, prec, device, preroll, warmup, steps, dist=None, keep=False, parity=None):
    """One of the other BASELINE configurations as a short lockstep leg (same workload family, pre-rolled).  parity = (quads, periods):
    the oracle check of parity_on_workload continued from the state the timed launches ended in (the f32 legs carry it: their
    claim is the 1e-4 budget ON this workload; the f64 legs' parity is covered by the GPU tests)."""
    refs_leg = (np.ascontiguousarray(refs[0][:B]), np.ascontiguousarray(refs[1][:B]))
    e, _cfg = make_engine(B, N, nb, prec, device, 0, 0, refs=refs_leg)
    n_sub = e.plant_substeps(0.01, 5e-3)
    e.sim_run(preroll, n_sub, 5e-3)
    dt, dt_own, k_avg = lockstep_leg(e, n_sub, warmup, steps, dist)
    its, status = e.get_qp_iter(), e.get_status()
    st = e.get_tracking_stats()
    groups = e.get_groups()
    # one period of the WHOLE batch for the byte rates: the step kernel's own launch time by HIP events with one group; with several groups
    # (a streaming batch: mpcq_tuning.groups automatic = 2) the events sit on group 0's launches, which cover B / groups quadrotors and share
    # the device with the other groups' -- there the period is the host clock over the call (order and plant launches included: it errs low)
    k_period = k_avg if groups == 1 else dt_own / steps
    out = {"config": name, "value": B * steps / dt_own, "unit": "control steps/s", "dtype": "f64" if prec == PRECISION_F64 else "f32",
           "batch": B, "horizon_nodes": N, "rgp_basis": nb, "steps": steps, "warmup": warmup, "preroll_periods": preroll,
           "ms_per_step": 1e3 * dt_own / steps, "kernel_avg_ms": 1e3 * k_avg, "groups": groups, "period_ms_for_byte_rates": 1e3 * k_period,
           "algorithmic_gbs": algorithmic_bytes(N, nb, 8 if prec == PRECISION_F64 else 4) * B / k_period / 1e9,
           "mean_qp_passes": float(qp_passes(its).mean()), "failed": int(((status & 7) != 0).sum()),
           "rms_pos_m": float(np.sqrt(st[0] / (3 * max(st[2], 1))))}
    if groups > 1:
        out["groups_note"] = (f"mpcq_sim_steps runs this batch as {groups} contiguous groups, each advancing in lockstep on a HIP stream of its own: the tail of one "
                              "group's launch (the device draining) is filled by the other's next launch; every quadrotor still advances `steps` periods "
                              "between the two barriers; kernel_avg_ms = group 0's launches (B / groups quadrotors, sharing the device)")
    if prec == PRECISION_F32:
        out["low_accuracy_last_step"] = int((status == 8).sum())      # MPCQ_SOLVE_LOW_ACCURACY: refinement not converged (expected 0)
    if parity is not None:
        r = parity_on_workload(dump_engine(e), refs_leg, N, nb, device, quads=parity[0], periods=parity[1], precision=prec)
        out["parity_on_workload"] = {k: r[k] for k in ("max_rel_dev", "max_rel_dev_per_quad", "rgp_max_rel_dev", "quad_steps", "fallbacks", "flip_marked",
                                                        "multi_pass", "failed", "saturated_controls", "quads", "periods", "mode")}
    if keep:
        return out, e, dt
    e.close()
    return out


def make_engine(B, N, nb, precision, device, first_index, seed, lib_path=None, periods=1000, refs=None, tune=None):
    """refs: (traj, lens) generated earlier (bench.workload forks worker processes: call it BEFORE this process touches the
    GPU and pass the result here)."""
    traj, lens = refs if refs is not None else workload(seed, first_index, B, periods)
    cfg = EngineConfig(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb),
                       theta=[1.0, 0.1, 0.1], dt_pred=0.01, device=device, precision=precision, tune=tune)
    e = Engine(cfg, lib_path=lib_path)
    e.set_trajectories(traj, lens)
    e.sim_reset(np.tile(X0, (B, 1)))
    return e, cfg


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical count."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(N, nb, seed, budget_s=18.0, refs=None, start=None):
    """The fp64 CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload: the first quadrotors of the GPU
    run, same references, continued closed loop from the state the GPU run ended in (`start`: iterate, RGP state, cursors
    and plant states after pre-roll + warm-up + timed steps, to within one control period -- the same stationary mix of
    flights, saturated quadrotors included; without `start` the sample begins at hover).  Swept over OpenMP team sizes; the best is `value`, the
    single-thread figure rides along.  Reported next to the GPU number; not the thing measured or shipped."""
    from oracle.oracle import OracleEngine
    logical, phys = os.cpu_count() or 1, physical_cores()
    try:
        native = True
        OracleEngine(EngineConfig(batch=1, N=5), native=True).close()
    except Exception:
        native = False
    quota = None
    try:      # cgroup v2 CPU quota of the container ("max 100000" = unlimited)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    # team sizes up to the CPUs this container may actually use (cgroup quota, affinity), one step beyond (2x) to show the
    # plateau; a third of the budget goes to the sweep, the rest to a longer sample at the best team size
    usable = int(min(logical, affinity, np.ceil(quota) if quota else logical))
    sweep = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, usable, 2 * usable) if 1 <= t <= min(2 * usable, logical, affinity)})
    runs = []

    def sample(threads, seconds, max_steps):
        B = max(32, 4 * threads)
        cfg = EngineConfig(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb),
                           theta=[1.0, 0.1, 0.1], dt_pred=0.01)
        o = OracleEngine(cfg, native=native)
        o.set_threads(threads)
        if refs is not None and refs[0].shape[0] >= B:      # the first B quadrotors of the GPU run's own references
            traj, lens = refs[0][:B], refs[1][:B]
        else:
            traj, lens = workload(seed, 0, B, 500)
        o.set_trajectories(traj, lens)
        from_start = start is not None and refs is not None and start["x"].shape[0] >= B
        if from_start:
            o.set_state(**{k: v[:B] for k, v in start["state"].items()})
            x = start["x"][:B].copy()
        else:
            x = np.tile(X0, (B, 1))
        for _ in range(2):
            w, _ = o.step(x)
            x = o.plant_control_period(x, w, 0.01, 5e-3)[0]
        steps, t_step = 0, 0.0
        t_end = time.perf_counter() + seconds
        while (time.perf_counter() < t_end or steps < 3) and steps < max_steps:
            t0 = time.perf_counter()
            w, _ = o.step(x)
            t_step += time.perf_counter() - t0
            x = o.plant_control_period(x, w, 0.01, 5e-3)[0]
            steps += 1
        o.set_threads(logical)
        o.close()
        return {"threads": threads, "quads": B, "steps": steps, "steps_per_s": B * steps / t_step,
                "us_per_step_per_thread": 1e6 * t_step * threads / (B * steps)}, from_start

    for threads in sweep:
        r, from_start = sample(threads, budget_s / 3 / len(sweep), 400)
        runs.append(r)
    best_threads = max(runs, key=lambda r: r["steps_per_s"])["threads"]
    long_run, from_start = sample(best_threads, 2 * budget_s / 3, 3000)
    long_run["long_sample"] = True
    runs.append(long_run)
    best = long_run      # the longer sample at the best team size of the sweep is the reported figure
    return {"value": best["steps_per_s"], "unit": "control steps/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['quads']} quads x {best['steps']} closed-loop steps "
                      + ("continued from the end state of the GPU run (same flights, same regime)" if from_start else "from hover")
                      + f", N={N} nb={nb}, fp64 C++ oracle "
                      f"(dense condensing + IPM, per-thread workspaces), OpenMP {best['threads']} threads, "
                      f"{'-march=native' if native else 'generic'} build",
            "single_thread_steps_per_s": runs[0]["steps_per_s"], "us_per_step_per_thread": best["us_per_step_per_thread"],
            "host": {"logical_cpus": logical, "physical_cores": phys, "affinity_cpus": affinity, "cgroup_cpu_quota": quota},
            "sweep": runs}


def call_with_timeout(fn, seconds=90.0):
    """Run fn() in a daemon thread; (True, result) or (False, exception / 'timeout').  A collective that never returns
    (a rank that died, a fabric problem) must not take the whole bench line with it."""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except Exception as ex:      # noqa: BLE001
            box["e"] = ex
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return False, "timeout"
    if "e" in box:
        return False, box["e"]
    return True, box.get("r")


SWARM_PER_RANK = 8192          # BASELINE configs[3]: 65 536 quadrotors over 8 GPUs
SWARM_WARM, SWARM_STEPS = 5, 20


def reduce_stats(e, dist, world, use_rccl):
    """The path's only collective: the 5-number tracking statistic, all-reduced by RCCL inside libmpcq.so (SUM of slots
    0,1,2,4, MAX of slot 3); over the host group when RCCL is not available (and the line says so)."""
    how, hung, stats = ("rccl" if use_rccl else "gloo"), False, None
    if world == 1:
        return e.get_tracking_stats(), "single", False
    import torch
    if use_rccl:
        oks, rs = call_with_timeout(e.allreduce_tracking_stats)
        flag = torch.tensor([1 if oks else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            stats = rs
        else:
            hung = rs == "timeout"
            how = "gloo (RCCL all-reduce failed)"
    if stats is None:
        st = e.get_tracking_stats()
        ssum = torch.tensor([st[0], st[1], st[2], 0.0, st[4]], dtype=torch.float64)
        smax = torch.tensor([st[3]], dtype=torch.float64)
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX)
        stats = ssum.numpy().copy()
        stats[3] = float(smax[0])
    return stats, how, hung


def per_rank_summary(dist, world, mine):
    """Every rank's own timing of a leg (dict of floats), gathered over the host group: {key: {min, mean, max}} + the list."""
    rows = [mine]
    if world > 1:
        rows = [None] * world
        dist.all_gather_object(rows, mine)
    out = {k: {"min": min(r[k] for r in rows), "mean": sum(r[k] for r in rows) / len(rows), "max": max(r[k] for r in rows)} for k in mine}
    out["ranks"] = rows
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0,
                    help="quadrotors per GPU; default 1024 (BASELINE configs[1]) at EVERY world size: the `value` series over N GPUs is a "
                         "weak-scaling series.  configs[3] (8192 per GPU, 65 536 over 8) rides in every line as the `swarm` leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configurations (and the swarm leg)")
    ap.add_argument("--swarm-per-rank", type=int, default=0,
                    help=f"quadrotors per GPU of the configs[3] `swarm` leg (default {SWARM_PER_RANK} = 65 536 / 8; a smaller value is a testing aid and "
                         "switches the leg on for any --batch)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity check on the workload")
    ap.add_argument("--steady", type=int, default=200, help="control periods of the steady-state leg that follows the timed region on the same engine "
                                                            "(one GPU only; 0: skip it and the other-seeds legs)")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--nb", type=int, default=10)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="arithmetic of the QP solve (state and QP data are always formed in double)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short secondary run in the other precision")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll", type=int, default=PREROLL,
                    help="un-timed control periods before the warm-up, the same for every --steps/--warmup: the timed region "
                         "starts in the stationary mix of a swarm in continuous operation (each quadrotor chains min-snap flights of "
                         "2.6 - 9.8 s; after 6 s their phases are spread), whatever --steps and --warmup are")
    ap.add_argument("--strict-rccl", dest="strict_rccl", action="store_true", default=None,
                    help="exit non-zero when WORLD_SIZE > 1 and the RCCL reduction did not run (the default under WORLD_SIZE > 1)")
    ap.add_argument("--no-strict-rccl", dest="strict_rccl", action="store_false",
                    help="accept a reduction over the host group when RCCL cannot be brought up (the line says which one ran)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("MPCQ_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # MPCQ_BENCH_DEVICE: testing aid (several ranks on one GPU)
    strict = (world > 1) if args.strict_rccl is None else args.strict_rccl
    dist = None
    if world > 1:
        import torch.distributed as dist  # host-side rendezvous only (barrier, max, id broadcast)
        dist.init_process_group("gloo", init_method="env://")
    if world != args.gpus and rank == 0:
        print(f"# note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)

    B = args.batch if args.batch > 0 else 1024
    N, nb = args.horizon, args.nb
    prec = PRECISION_F64 if args.precision == "f64" else PRECISION_F32
    itemsize = 8 if prec == PRECISION_F64 else 4
    STEADY = args.steady if world == 1 else 0          # further periods on the same engine behind the timed region
    LAT = 20 if STEADY else 0                          # lockstep periods of the latency_roofline leg (one call each, counters read back)
    periods = args.preroll + args.warmup + args.steps + STEADY + LAT
    headline = (B, N, nb) == (1024, 20, 10)
    legs = world == 1 and not args.no_configs and headline
    swarm = not args.no_configs and (headline or args.swarm_per_rank > 0)   # configs[3]: 8192 quadrotors on every rank, at every world size
    SWARM = args.swarm_per_rank if args.swarm_per_rank > 0 else SWARM_PER_RANK
    seeds_alt = [1, 2, 3] if (world == 1 and headline and STEADY > 0) else []
    t_gen = time.perf_counter()
    refs = workload(args.seed, rank * B, B, periods)      # host-side generation (worker processes) before the GPU is touched
    CFG_PRE, CFG_WARM, CFG_STEPS = 300, 5, 20
    refs_cfg = workload(args.seed, 0, 8192, CFG_PRE + CFG_WARM + CFG_STEPS) if legs else None   # shared by the legs (150 rows >= N skip)
    refs_swarm = workload(args.seed, rank * SWARM, SWARM, args.preroll + SWARM_WARM + SWARM_STEPS) if swarm else None
    refs_seeds = {sd: workload(sd, 0, B, args.preroll + 10 + 50) for sd in seeds_alt}
    t_gen = time.perf_counter() - t_gen
    e, cfg = make_engine(B, N, nb, prec, local_rank, rank * B, args.seed, periods=periods, refs=refs)
    stats_reduce = "single"
    rccl_hung = False
    uid_bytes = None
    if world > 1:
        # the only collective of the path: RCCL all-reduce of the 5-number swarm statistic inside libmpcq.so.
        # If RCCL cannot be brought up on this node the statistic is reduced over the host group instead
        # (and the bench line says so; with --strict-rccl, the default here, the run then fails); the timed region has no collective.
        stats_reduce = "rccl"
        okid, rid = call_with_timeout(lambda: e.comm_unique_id() if rank == 0 else None)
        uid = [rid if okid else f"ERR {rid}"]
        dist.broadcast_object_list(uid, src=0)
        ok = 1
        if isinstance(uid[0], (bytes, bytearray)):
            okc, rc = call_with_timeout(lambda: e.comm_init(rank, world, uid[0]))
            if not okc:
                ok = 0
                rccl_hung = rc == "timeout"
                print(f"# rank {rank}: RCCL init failed: {rc}", file=sys.stderr)
        else:
            ok = 0
        import torch
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            stats_reduce = "gloo (RCCL unavailable)"

    n_sub = e.plant_substeps(0.01, 5e-3)     # 100 Hz odometry = 2 plant substeps of 5 ms (the reference's float-accumulated loop)
    if args.preroll > 0:
        # one persistent launch (every quadrotor advances through the pre-roll on its own; bit-identical to per-period
        # launches): the lockstep kernel's launches seen by a profiler are then exactly warm-up + timed steps
        e.sim_run(args.preroll, n_sub, 5e-3)
    elapsed, own_elapsed, k_avg_own = lockstep_leg(e, n_sub, args.warmup, args.steps, dist)   # W untimed launches, barrier, K timed launches, barrier
    start = None
    if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_parity):   # where the CPU legs continue from
        start = dump_engine(e)
    ktime, klaunch = e.get_kernel_time()     # HIP events around the step_kernel launches of the timed region (every 4th; every launch when steps < 8)
    kmin, kmax = e.get_kernel_time_minmax()
    its = e.get_qp_iter()
    status = e.get_status()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    # every rank's own clock around its K launches (the barrier that ends the region is included in `elapsed`, not here)
    ranks = per_rank_summary(dist, world, {"ms_per_step": 1e3 * own_elapsed / args.steps, "kernel_avg_ms": 1e3 * k_avg_own,
                                           "steps_per_s": B * args.steps / own_elapsed,
                                           "host_generation_s": t_gen})      # every rank generates its own references concurrently (round-5 verdict, item 9)
    stats, stats_reduce2, hung2 = reduce_stats(e, dist, world, stats_reduce == "rccl")
    if world > 1:
        stats_reduce, rccl_hung = (stats_reduce2 if stats_reduce == "rccl" else stats_reduce), rccl_hung or hung2
    # ---- steady state: the driver's --steps 20 window is 6 ms of a workload whose launch time follows the number of saturated
    #      quadrotors in flight; 200 further periods on the same engine give the representative figure
    steady = None
    if STEADY:
        dt_s, _own, k_s = lockstep_leg(e, n_sub, 0, STEADY)
        steady = {"value": B * STEADY / dt_s, "unit": "control steps/s", "steps": STEADY, "ms_per_step": 1e3 * dt_s / STEADY, "kernel_avg_ms": 1e3 * k_s,
                  "note": f"the {STEADY} control periods that follow the timed region, same engine: the representative lockstep rate of this workload "
                          "(`value` above is the driver's window; its launches hold more or fewer saturated quadrotors by chance)"}

    lat = latency_roofline(e, n_sub, N, LAT) if LAT else None
    if lat is None:
        LAT = 0
    # ---- configs[3]-shaped leg on every rank: 8192 quadrotors per GPU (65 536 over 8), same pre-roll as the headline
    swarm_out = None
    if swarm:
        leg, es, agg_dt = config_leg(f"configs[3] per rank: batch {SWARM} of {SWARM * 8}, N=20, RGP 10 basis pts", refs_swarm, SWARM, 20, 10,
                                     PRECISION_F64, local_rank, args.preroll, SWARM_WARM, SWARM_STEPS, dist=dist, keep=True)
        if dist is not None:      # between the barriers, MAX over ranks (as the headline)
            import torch
            t = torch.tensor([agg_dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            agg_dt = float(t[0])
        sranks = per_rank_summary(dist, world, {"ms_per_step": leg["ms_per_step"], "kernel_avg_ms": leg["kernel_avg_ms"], "steps_per_s": leg["value"]})
        # configs[3] IS "RCCL reduce of tracking RMSE": this engine reduces its own statistic over the rank's communicator (the one the
        # headline engine initialised: one communicator per rank, mpcq_comm_share)
        swarm_rccl = world > 1 and stats_reduce == "rccl"
        if swarm_rccl:
            try:
                es.comm_share(e)
            except Exception as ex:      # noqa: BLE001
                print(f"# rank {rank}: mpcq_comm_share failed: {ex}", file=sys.stderr)
                swarm_rccl = False
            import torch
            flag = torch.tensor([1 if swarm_rccl else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            swarm_rccl = int(flag[0]) == 1
        sstats, show, hung3 = reduce_stats(es, dist, world, swarm_rccl)
        rccl_hung = rccl_hung or hung3
        if world > 1 and stats_reduce == "rccl" and show != "rccl":
            stats_reduce = "rccl (headline), " + show + " (swarm)"      # strict mode: the run fails, the line says which reduction did not run
        es.close()
        best = sranks["steps_per_s"]["max"]
        swarm_out = dict(leg)
        # HBM roofline of this leg (per GPU): algorithmic bytes over the launch time by HIP events; memory-side traffic from the rocprofv3 passes of this
        # configuration, attached only when they were taken on this build (source hash)
        sw_bytes = algorithmic_bytes(20, 10, 8) * SWARM
        sw_period_ms = leg["period_ms_for_byte_rates"]      # (= kernel_avg_ms with one group; the period of the whole batch with several)
        sw_roof = {"bound": "hbm", "achieved": sw_bytes / (sw_period_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                   "algorithmic_bytes_per_launch": sw_bytes, "period_ms": sw_period_ms, "groups": leg["groups"]}
        sw_roof["frac"] = sw_roof["achieved"] / HBM_PEAK_GBS
        import glob as _glob
        for tf in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_swarm_b8192.json"))):      # the profile taken on THIS build, if any
            try:
                with open(tf) as fh:
                    tsw = json.load(fh)
            except (OSError, ValueError):
                continue
            if tsw.get("source_sha16") == kernel_source_sha16():
                sw_roof["traffic"] = tsw["hbm_bytes_per_launch"]
                sw_roof["traffic_frac_of_peak"] = tsw["hbm_bytes_per_launch"] / (sw_period_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                sw_roof["traffic_source"] = f"profiles/{os.path.basename(tf)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tools/large_batch.py 8192 20 10, same build)"
        swarm_out["roofline"] = sw_roof
        swarm_out.update({"config": f"BASELINE configs[3]: swarm of {SWARM * world} quadrotors, {SWARM} per GPU over {world} GPU(s)"
                                    + (" (= the per-rank shard of the 65 536-quadrotor swarm)" if world == 1 else ""),
                          "value": SWARM * world * SWARM_STEPS / agg_dt, "n_gpus": world, "global_batch": SWARM * world,
                          "ms_per_step": 1e3 * agg_dt / SWARM_STEPS, "per_rank": sranks, "efficiency_vs_best_rank": SWARM * world * SWARM_STEPS / agg_dt / (world * best),
                          "stats_reduce": show, "rms_pos_m": float(np.sqrt(sstats[0] / (3 * max(sstats[2], 1)))), "tracking_steps": float(sstats[2])})

    if rank == 0:
        total_steps = B * world * args.steps
        value = total_steps / elapsed
        k_avg = ktime / max(klaunch, 1)
        bytes_launch = algorithmic_bytes(N, nb, itemsize) * B
        achieved = bytes_launch / k_avg / 1e9
        flops_launch = algorithmic_flops(N, nb, float(qp_passes(its).mean())) * B
        out = {
            "metric": "batched MPC+RGP control steps/sec (N=20 horizon)" if N == 20 else f"batched MPC+RGP control steps/sec (N={N} horizon)",
            "value": value, "unit": "control steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] per GPU: batch {B} hummingbird quadrotors" + (f" on each of {world} GPUs ({B * world} in total)" if world > 1 else "")
                                   + f", N={N}, RGP {nb} basis pts/axis, closed loop with on-device drag plant, continuous operation on seeded "
                                     "random-waypoint min-snap flights (3 waypoints each, v_max=a_max=12)",
                       "batch_per_gpu": B, "global_batch": B * world, "horizon_nodes": N, "rgp_basis": nb, "preroll_periods": args.preroll,
                       "parallelism": f"shard{world}" if world > 1 else "single", "threads_per_quad": 64,
                       "stats_reduce": stats_reduce, "rccl_ok": world == 1 or stats_reduce == "rccl",
                       "scaling_note": f"weak scaling: {B} quadrotors per GPU at every world size, sharded by global index, no collective on the data path; "
                                       "BASELINE configs[3] (8192 per GPU, 65 536 over 8 GPUs) is the `swarm` object of every line, at the same "
                                       "per-GPU batch for every world size as well"},
            "per_rank": ranks,
            "efficiency_vs_best_rank": value / (world * ranks["steps_per_s"]["max"]),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": f"mpcq::step_kernel<{'double' if prec == PRECISION_F64 else 'float'}>",
                         "kernel_avg_ms": 1e3 * k_avg, "kernel_min_ms": 1e3 * kmin, "kernel_max_ms": 1e3 * kmax,
                         "kernel_launches": args.steps, "kernel_launches_timed": klaunch,
                         "timing": "HIP events on the engine's stream around the step-kernel launches of the timed region "
                                   "(every 4th launch; every launch when steps < 8: an event pair between dependent launches costs dispatch overlap, 2.4 % at 20 launches)",
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "path is latency/VALU/LDS bound, not HBM bound (DESIGN.md): secondary figure below",
                         "vector_flops": {"achieved_tflops": flops_launch / k_avg / 1e12,
                                          "peak_tflops": FP_VECTOR_PEAK[args.precision],
                                          "frac": flops_launch / k_avg / 1e12 / FP_VECTOR_PEAK[args.precision]}},
            "solver": {"mean_qp_passes": float(qp_passes(its).mean()), "max_qp_passes": int(qp_passes(its).max()),
                       "ipm_fallbacks_last_step": int(qp_fallback(its).sum()), "failed": int(((status & 7) != 0).sum())},
            "host_generation_s": t_gen,
            "tracking": {"rms_pos_m": float(np.sqrt(stats[0] / (3 * max(stats[2], 1)))), "steps": float(stats[2]),
                         "max_pos_err_m": float(np.sqrt(stats[3])), "failed_instances": float(stats[4])},
        }
        if steady is not None:
            out["steady_state"] = steady
            # machine-readable form of "which number is representative": `value` is the driver's short window, whose launches hold more or
            # fewer saturated quadrotors by chance; the steady-state leg is the figure to quote
            out["value_window"] = {"steps": args.steps, "ms": 1e3 * elapsed, "representative": "steady_state.value",
                                   "value_over_steady_state": value / steady["value"]}
        if lat is not None:
            out["latency_roofline"] = lat
        if swarm_out is not None:
            out["swarm"] = swarm_out
        # HBM bytes per launch from rocprofv3 --pmc passes (profiles/README.md): attached only when the profile was taken on
        # THIS build (source hash) with THIS command line; otherwise null -- a number from another build is not this run's traffic.
        import glob
        sha = kernel_source_sha16()
        from mpc_quad_ros_amd import _lib as _l
        ver = _l.load().mpcq_version().decode()
        out["library"] = {"version": ver, "built_from_these_sources": sha in ver}      # the Makefile puts the source hash into the version string
        key = {"steps": args.steps, "warmup": args.warmup, "preroll": args.preroll, "seed": args.seed, "batch": B, "N": N, "nb": nb,
               "precision": args.precision}
        out["roofline"]["traffic_note"] = f"no PMC profile of build {sha} for this command line under profiles/"
        near = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_*.json"))):
            try:
                with open(f) as fh:
                    t = json.load(fh)
            except (OSError, ValueError):
                continue
            if t.get("source_sha16") == sha and t.get("args") == key:
                out["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = {"file": os.path.basename(f), "source_sha16": sha, "command": t.get("command"),
                                                     "commit": t.get("commit")}
                out["roofline"]["traffic_note"] = t.get("note", "")
                near = None
                break
            # same build, same workload, other --steps / --warmup: the bytes per launch barely depend on them (104.7 MB at 20 / 5, 106.3 MB at
            # 200 / 20); attached with the command line it was taken on when nothing matches exactly
            same = {k: v for k, v in (t.get("args") or {}).items() if k not in ("steps", "warmup")} == {k: v for k, v in key.items() if k not in ("steps", "warmup")}
            if t.get("source_sha16") == sha and same and (near is None or abs(t["args"]["steps"] - args.steps) < abs(near[1]["args"]["steps"] - args.steps)):
                near = (f, t)
        if near is not None:
            f, t = near
            out["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = {"file": os.path.basename(f), "source_sha16": sha, "command": t.get("command"), "commit": t.get("commit"),
                                                 "other_window": f"taken with --steps {t['args']['steps']} --warmup {t['args']['warmup']} (this run: {args.steps} / {args.warmup}), same build and workload"}
            out["roofline"]["traffic_note"] = t.get("note", "")
        if seeds_alt:
            # the same workload from other seeds (the solver heuristics were tuned on the default one): 50 periods each
            out["seeds"] = []
            for sd in seeds_alt:
                es, _ = make_engine(B, N, nb, prec, local_rank, 0, sd, refs=refs_seeds[sd])
                es.sim_run(args.preroll, n_sub, 5e-3)
                dt_s, _own, k_s = lockstep_leg(es, n_sub, 10, 50)
                st_s = es.get_status()
                out["seeds"].append({"seed": sd, "value": B * 50 / dt_s, "steps": 50, "warmup": 10, "kernel_avg_ms": 1e3 * k_s, "failed": int(((st_s & 7) != 0).sum())})
                es.close()
        if world == 1 and not args.no_alt:
            # Same K periods as ONE launch in which every quadrotor runs through its periods without waiting for the
            # slowest member of the batch (mpcq_sim_run): the lockstep figure above is what a controller fed by live
            # measurements gets per tick, this one is the capacity of the device as a closed-loop swarm simulator.
            x_lock, w_lock = e.sim_get_state()
            e.close()
            e3, _ = make_engine(B, N, nb, prec, local_rank, rank * B, args.seed, periods=periods, refs=refs)
            e3.sim_run(args.preroll + args.warmup, n_sub, 5e-3)
            e3.synchronize()
            ta = time.perf_counter()
            e3.sim_run(args.steps + STEADY + LAT, n_sub, 5e-3)
            e3.synchronize()
            tb = time.perf_counter()
            k3, _l3 = e3.get_kernel_time()
            x_run, w_run = e3.sim_get_state()
            KF = args.steps + STEADY + LAT
            out["free_running"] = {"value": B * KF / (tb - ta), "unit": "control steps/s", "dtype": args.precision, "steps": KF,
                                   "ms_per_step": 1e3 * (tb - ta) / KF, "kernel_ms_per_step": 1e3 * k3 / KF,
                                   "launches": 1, "bitwise_equal_to_lockstep": bool(np.array_equal(x_run, x_lock) and np.array_equal(w_run, w_lock)),
                                   "note": "one persistent launch over the timed AND the steady-state periods, each workgroup advances its quadrotor through all of them "
                                           "(step + plant) on its own; identical arithmetic and results, no per-period wait for the "
                                           "slowest instance of the batch"}
            e3.close()
            # The same periods as lockstep launches of GROUPS of the batch (mpcq_tuning.groups): each group of B / G quadrotors advances in lockstep on
            # its own stream, so a launch waits for the slowest quadrotor of its group only and the groups fill each other's waits.  Between
            # `value` (one launch per period over the whole batch: what a controller fed by live measurements gets per tick) and free_running.
            GL = 4
            e4, _ = make_engine(B, N, nb, prec, local_rank, rank * B, args.seed, periods=periods, refs=refs, tune=dict(groups=GL))
            e4.sim_run(args.preroll + args.warmup, n_sub, 5e-3)
            e4.synchronize()
            ta = time.perf_counter()
            e4.sim_steps(KF, n_sub, 5e-3)
            e4.synchronize()
            tb = time.perf_counter()
            x_g, w_g = e4.sim_get_state()
            out["lockstep_groups"] = {"value": B * KF / (tb - ta), "unit": "control steps/s", "dtype": args.precision, "steps": KF, "groups": e4.get_groups(),
                                      "ms_per_step": 1e3 * (tb - ta) / KF, "bitwise_equal_to_lockstep": bool(np.array_equal(x_g, x_lock) and np.array_equal(w_g, w_lock)),
                                      "note": f"mpcq_sim_steps with mpcq_tuning.groups = {GL}: the batch as {GL} contiguous groups of {B // GL}, each advancing in lockstep "
                                              "(one launch per period and group) on a HIP stream of its own; one call, every quadrotor advances `steps` periods; identical "
                                              "results.  Not the headline: `value` keeps one launch per period over the whole batch"}
            e4.close()
            alt = "f32" if args.precision == "f64" else "f64"
            e2, _ = make_engine(B, N, nb, PRECISION_F32 if alt == "f32" else PRECISION_F64, local_rank, 0, args.seed, periods=periods, refs=refs)
            e2.sim_steps(args.preroll + args.warmup, n_sub, 5e-3)
            e2.synchronize()
            ta = time.perf_counter()
            e2.sim_steps(args.steps, n_sub, 5e-3)
            e2.synchronize()
            tb = time.perf_counter()
            k2, l2 = e2.get_kernel_time()
            st2 = e2.get_status()
            leg2 = {"dtype": alt, "value": B * args.steps / (tb - ta), "unit": "control steps/s",
                    "kernel_avg_ms": 1e3 * k2 / max(l2, 1), "failed": int(((st2 & 7) != 0).sum()),
                    "note": "same workload, same launches, the other precision.  f32 = mixed precision (include/mpcq.h): float stage records, float Riccati "
                            "factorisation on the matrix cores, QP solution refined against fp64 residuals; f64 = the reference's own arithmetic"}
            if alt == "f32":
                leg2["low_accuracy_last_step"] = int((st2 == 8).sum())      # MPCQ_SOLVE_LOW_ACCURACY: refinement not converged (expected 0)
                if not args.no_parity:
                    leg2["parity_on_workload"] = parity_on_workload(dump_engine(e2), refs, N, nb, local_rank, quads=64, periods=30, precision=PRECISION_F32)
            out["f32" if alt == "f32" else "alt_precision"] = leg2
            e2.close()
        if legs:
            # the other BASELINE configurations, reachable from the driver's command: short lockstep legs after the headline
            par = None if args.no_parity else (32, 10)
            out["configs"] = [
                config_leg("configs[2]: batch 8192, N=20, RGP 20 basis pts", refs_cfg, 8192, 20, 20, PRECISION_F64, local_rank, CFG_PRE, CFG_WARM, CFG_STEPS),
                config_leg("configs[2]: batch 8192, N=20, RGP 20 basis pts", refs_cfg, 8192, 20, 20, PRECISION_F32, local_rank, CFG_PRE, CFG_WARM, CFG_STEPS, parity=par),
                {k: v for k, v in swarm_out.items() if k not in ("per_rank", "efficiency_vs_best_rank", "n_gpus", "global_batch", "stats_reduce", "tracking_steps", "roofline")}
                | {"config": f"configs[3] per rank: batch {SWARM} of {SWARM * 8}, N=20, RGP 10 basis pts (the `swarm` leg of this line: same pre-roll as an N > 1 run)"},
                config_leg(f"configs[3] per rank: batch {SWARM} of {SWARM * 8}, N=20, RGP 10 basis pts", refs_swarm, SWARM, 20, 10, PRECISION_F32, local_rank,
                           args.preroll, SWARM_WARM, SWARM_STEPS, parity=par),
                config_leg("configs[4]: batch 4096, N=50, RGP 50 basis pts", refs_cfg, 4096, 50, 50, PRECISION_F64, local_rank, CFG_PRE, CFG_WARM, CFG_STEPS),
                config_leg("configs[4]: batch 4096, N=50, RGP 50 basis pts", refs_cfg, 4096, 50, 50, PRECISION_F32, local_rank, CFG_PRE, CFG_WARM, CFG_STEPS, parity=par),
            ]
        if world == 1 and not args.no_parity:
            out["parity_on_workload"] = parity_on_workload(start, refs, N, nb, local_rank, quads=64, periods=30)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, nb, args.seed, refs=refs, start=start)
        print(json.dumps(out))
    rccl_ok = world == 1 or stats_reduce == "rccl"
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if strict and not rccl_ok and not rccl_hung:
            sys.exit(3)
        if rccl_hung:            # a thread is still blocked inside RCCL: skip the destructors
            _flush_line()        # (os._exit skips the `finally` of _main_with_clean_stdout)
            os._exit(3 if strict else 0)


_REAL_STDOUT = None   # (fd, buffer) while main() runs under _main_with_clean_stdout


def _flush_line():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        return
    fd, buf = _REAL_STDOUT
    _REAL_STDOUT = None
    text = buf.getvalue()
    if text:
        os.write(fd, text.encode())
    os.close(fd)


def _main_with_clean_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (under torch.distributed.run the Gloo and RCCL banners of
    rank 0 arrive on fd 1): while the benchmark runs fd 1 points at stderr, and the line -- everything this program itself prints to
    sys.stdout -- is written to the real stdout at the end."""
    import io
    global _REAL_STDOUT
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    py_stdout = sys.stdout
    _REAL_STDOUT = (real, io.StringIO())
    sys.stdout = _REAL_STDOUT[1]
    try:
        main()
    finally:
        sys.stdout = py_stdout
        _flush_line()


if __name__ == "__main__":
    _main_with_clean_stdout()
