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

from mpc_quad_ros_amd.engine import Engine  # noqa: E402
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


def workload(seed, first_index, B, periods):
    """Continuous operation of the node, per quadrotor: min-snap flights through 3 random waypoints (v_max = a_max = 12, the
    launch defaults), each requested from the end point of the previous one (mpc_quad_ros_amd.trajectories.minsnap_mission),
    long enough for `periods` control periods plus one horizon."""
    return swarm_missions(seed, first_index, B, periods + 150, v_max=12.0, a_max=12.0)


def make_engine(B, N, nb, precision, device, first_index, seed, lib_path=None, periods=1000, refs=None):
    """refs: (traj, lens) generated earlier -- the generator forks worker processes, which is done BEFORE this process
    touches the GPU."""
    traj, lens = refs if refs is not None else workload(seed, first_index, B, periods)
    cfg = EngineConfig(batch=B, N=N, T=1.0, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb),
                       theta=[1.0, 0.1, 0.1], dt_pred=0.01, device=device, precision=precision)
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
    sweep = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, phys, logical) if 1 <= t <= min(logical, affinity)})
    runs = []
    for threads in sweep:
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
        t_end = time.perf_counter() + budget_s / len(sweep)
        while (time.perf_counter() < t_end or steps < 3) and steps < 400:
            t0 = time.perf_counter()
            w, _ = o.step(x)
            t_step += time.perf_counter() - t0
            x = o.plant_control_period(x, w, 0.01, 5e-3)[0]
            steps += 1
        runs.append({"threads": threads, "quads": B, "steps": steps, "steps_per_s": B * steps / t_step,
                     "us_per_step_per_thread": 1e6 * t_step * threads / (B * steps)})
        o.set_threads(logical)
        o.close()
    best = max(runs, key=lambda r: r["steps_per_s"])
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="quadrotors per GPU")
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
    ap.add_argument("--strict-rccl", action="store_true", help="exit non-zero when WORLD_SIZE > 1 and the RCCL reduction did not run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("MPCQ_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # MPCQ_BENCH_DEVICE: testing aid (several ranks on one GPU)
    dist = None
    if world > 1:
        import torch.distributed as dist  # host-side rendezvous only (barrier, max, id broadcast)
        dist.init_process_group("gloo", init_method="env://")
    if world != args.gpus and rank == 0:
        print(f"# note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)

    B, N, nb = args.batch, args.horizon, args.nb
    prec = PRECISION_F64 if args.precision == "f64" else PRECISION_F32
    itemsize = 8 if prec == PRECISION_F64 else 4
    periods = args.preroll + args.warmup + args.steps
    refs = workload(args.seed, rank * B, B, periods)      # host-side generation (worker processes) before the GPU is touched
    e, cfg = make_engine(B, N, nb, prec, local_rank, rank * B, args.seed, periods=periods, refs=refs)
    stats_reduce = "single"
    rccl_hung = False
    if world > 1:
        # the only collective of the path: RCCL all-reduce of the 5-number swarm statistic inside libmpcq.so.
        # If RCCL cannot be brought up on this node the statistic is reduced over the host group instead
        # (and the bench line says so); the timed region has no collective either way.
        stats_reduce = "rccl"
        rccl_hung = False
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

    def barrier():
        e.lib.mpcq_synchronize(e.h)          # hipStreamSynchronize on the engine's stream (the only one used)
        if dist is not None:
            dist.barrier()

    n_sub = e.plant_substeps(0.01, 5e-3)     # 100 Hz odometry = 2 plant substeps of 5 ms (the reference's float-accumulated loop)
    if args.preroll > 0:
        # one persistent launch (every quadrotor advances through the pre-roll on its own; bit-identical to per-period
        # launches): the lockstep kernel's launches seen by a profiler are then exactly warm-up + timed steps
        e.sim_run(args.preroll, n_sub, 5e-3)
    e.sim_steps(args.warmup, n_sub, 5e-3)
    barrier()
    t0 = time.perf_counter()
    e.sim_steps(args.steps, n_sub, 5e-3)     # K fused steps + plant, back-to-back on one stream
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    start = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # where the CPU baseline continues from
        start = {"state": e.get_state(), "x": e.sim_get_state()[0]}
    ktime, klaunch = e.get_kernel_time()     # HIP events around the step_kernel launches of the timed region (all of them when steps <= 50, else every 4th)
    kmin, kmax = e.get_kernel_time_minmax()
    its = e.get_qp_iter()
    status = e.get_status()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    stats = None
    if world > 1 and stats_reduce == "rccl":
        oks, rs = call_with_timeout(e.allreduce_tracking_stats)
        import torch
        flag = torch.tensor([1 if oks else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            stats = rs
        else:
            rccl_hung = rccl_hung or rs == "timeout"
            stats_reduce = "gloo (RCCL all-reduce failed)"
    if stats is None:
        stats = e.get_tracking_stats()
        if world > 1:
            import torch
            ssum = torch.tensor([stats[0], stats[1], stats[2], 0.0, stats[4]], dtype=torch.float64)
            smax = torch.tensor([stats[3]], dtype=torch.float64)
            dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
            dist.all_reduce(smax, op=dist.ReduceOp.MAX)
            stats = ssum.numpy().copy()
            stats[3] = float(smax[0])

    if rank == 0:
        total_steps = B * world * args.steps
        value = total_steps / elapsed
        k_avg = ktime / max(klaunch, 1)
        bytes_launch = algorithmic_bytes(N, nb, itemsize) * B
        achieved = bytes_launch / k_avg / 1e9
        flops_launch = algorithmic_flops(N, nb, float((its % 1000).mean())) * B
        out = {
            "metric": "batched MPC+RGP control steps/sec (N=20 horizon)" if N == 20 else f"batched MPC+RGP control steps/sec (N={N} horizon)",
            "value": value, "unit": "control steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] per GPU: batch {B} hummingbird quadrotors, N={N}, RGP {nb} basis pts/axis, "
                                   "closed loop with on-device drag plant, continuous operation on seeded random-waypoint min-snap flights (3 waypoints each, v_max=a_max=12)",
                       "batch_per_gpu": B, "global_batch": B * world, "horizon_nodes": N, "rgp_basis": nb, "preroll_periods": args.preroll,
                       "parallelism": f"shard{world}" if world > 1 else "single", "threads_per_quad": 64,
                       "stats_reduce": stats_reduce, "rccl_ok": world == 1 or stats_reduce == "rccl"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": f"mpcq::step_kernel<{'double' if prec == PRECISION_F64 else 'float'}>",
                         "kernel_avg_ms": 1e3 * k_avg, "kernel_min_ms": 1e3 * kmin, "kernel_max_ms": 1e3 * kmax,
                         "kernel_launches": args.steps, "kernel_launches_timed": klaunch,
                         "timing": "HIP events on the engine's stream around the step-kernel launches of the timed region "
                                   "(every launch when steps <= 50, else every 4th; MPCQ_KEV_STRIDE)",
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "path is latency/VALU/LDS bound, not HBM bound (DESIGN.md): secondary figure below",
                         "vector_flops": {"achieved_tflops": flops_launch / k_avg / 1e12,
                                          "peak_tflops": FP_VECTOR_PEAK[args.precision],
                                          "frac": flops_launch / k_avg / 1e12 / FP_VECTOR_PEAK[args.precision]}},
            "solver": {"mean_qp_passes": float((its % 1000).mean()), "max_qp_passes": int((its % 1000).max()),
                       "ipm_fallbacks_last_step": int((its >= 1000).sum()), "failed": int((status != 0).sum())},
            "tracking": {"rms_pos_m": float(np.sqrt(stats[0] / (3 * max(stats[2], 1)))), "steps": float(stats[2]),
                         "max_pos_err_m": float(np.sqrt(stats[3])), "failed_instances": float(stats[4])},
        }
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{args.precision}.json")))
        pmc = cands[-1] if cands and (B, N, nb) == (1024, 20, 10) else ""
        if pmc and os.path.exists(pmc):   # HBM bytes per launch from rocprofv3 --pmc passes of this same command (profiles/README.md)
            with open(pmc) as f:
                t = json.load(f)
            out["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = t.get("source")
            if args.precision == "f64":
                out["roofline"]["traffic_note"] = ("fp64 keeps the per-stage records (38.6 KB per instance) in global memory, streamed through L2, so "
                                                   "that 4 instead of 2 instances fit a CU; every further working set / interior-point iteration of a "
                                                   "quadrotor re-reads them, so the traffic grows with the pass count of the launch (DESIGN.md section 3.1)")
        if world == 1 and not args.no_alt:
            # Same K periods as ONE launch in which every quadrotor runs through its periods without waiting for the
            # slowest member of the batch (mpcq_sim_run): the lockstep figure above is what a controller fed by live
            # measurements gets per tick, this one is the capacity of the device as a closed-loop swarm simulator.
            x_lock, w_lock = e.sim_get_state()
            e.close()
            e3, _ = make_engine(B, N, nb, prec, local_rank, 0, args.seed, periods=periods, refs=refs)
            e3.sim_run(args.preroll + args.warmup, n_sub, 5e-3)
            e3.lib.mpcq_synchronize(e3.h)
            ta = time.perf_counter()
            e3.sim_run(args.steps, n_sub, 5e-3)
            e3.lib.mpcq_synchronize(e3.h)
            tb = time.perf_counter()
            k3, _l3 = e3.get_kernel_time()
            x_run, w_run = e3.sim_get_state()
            out["free_running"] = {"value": B * args.steps / (tb - ta), "unit": "control steps/s", "dtype": args.precision,
                                   "ms_per_step": 1e3 * (tb - ta) / args.steps, "kernel_ms_per_step": 1e3 * k3 / args.steps,
                                   "launches": 1, "bitwise_equal_to_lockstep": bool(np.array_equal(x_run, x_lock) and np.array_equal(w_run, w_lock)),
                                   "note": "one persistent launch, each workgroup advances its quadrotor through all K control periods "
                                           "(step + plant) on its own; identical arithmetic and results, no per-period wait for the "
                                           "slowest instance of the batch"}
            e3.close()
            alt = "f32" if args.precision == "f64" else "f64"
            e2, _ = make_engine(B, N, nb, PRECISION_F32 if alt == "f32" else PRECISION_F64, local_rank, 0, args.seed, periods=periods, refs=refs)
            e2.sim_steps(args.preroll + args.warmup, n_sub, 5e-3)
            e2.lib.mpcq_synchronize(e2.h)
            ta = time.perf_counter()
            e2.sim_steps(args.steps, n_sub, 5e-3)
            e2.lib.mpcq_synchronize(e2.h)
            tb = time.perf_counter()
            k2, l2 = e2.get_kernel_time()
            out["alt_precision"] = {"dtype": alt, "value": B * args.steps / (tb - ta), "unit": "control steps/s",
                                    "kernel_avg_ms": 1e3 * k2 / max(l2, 1),
                                    "note": "same workload with the QP arithmetic in the other precision.  f32 (all-LDS working set): relative control "
                                            "deviation vs the fp64 oracle <= 1e-4 on every warm-started solve of the six reference logs (median 2e-6); "
                                            "interior-point fallback solves reach 1.2e-4..1.5e-4 on two logs and 1.2e-3 on one tumbling step "
                                            "(profiles/r2_f32_log_report.json, DESIGN.md section 5).  f64 (stage records in global memory): <= 3e-10.  "
                                            "Both place 4 quadrotors per CU"}
            e2.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, nb, args.seed, refs=refs, start=start)
        print(json.dumps(out))
    rccl_ok = world == 1 or stats_reduce == "rccl"
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if args.strict_rccl and not rccl_ok and not rccl_hung:
            sys.exit(3)
        if rccl_hung:            # a thread is still blocked inside RCCL: skip the destructors
            sys.stdout.flush()
            os._exit(3 if args.strict_rccl else 0)


if __name__ == "__main__":
    main()
