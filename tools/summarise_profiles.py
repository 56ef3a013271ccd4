#!/usr/bin/env python3
"""Condense the rocprofv3 outputs gathered by tools/collect_profiles.sh (gpurun_out/) into the tracked files
under profiles/: <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc.json and r1_pmc_traffic_<prec>.json
(the per-launch HBM bytes bench.py reports as roofline.traffic).   usage: summarise_profiles.py [TAG]"""
import csv
import re
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r3_final"
ROUND = TAG.split("_")[0]     # r2_final -> r2: prefix of the traffic file bench.py picks up
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def is_lockstep(name):
    """step_kernel<Cfg<T, GAB, N, NB, RUN, GK>, ...>: RUN = false is the one-launch-per-period instance."""
    m = re.search(r"step_kernel<mpcq::Cfg<\w+, (?:true|false), -?\d+, -?\d+, (true|false)(?:, (?:true|false))?>", name)
    return bool(m) and m.group(1) == "false"


def one(pattern):
    g = sorted(glob.glob(os.path.join(O, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None   # newest run of that tag


bench = json.loads(open(os.path.join(O, f"bench_{TAG}.json")).read().strip().splitlines()[-1])
with open(os.path.join(P, f"{TAG}_bench.json"), "w") as f:
    json.dump(bench, f, indent=1)
ks = one(f"prof_{TAG}/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(P, f"{TAG}_kernel_stats.csv"))
    for r in csv.DictReader(open(ks)):
        if "step_kernel" in r["Name"]:
            print("kernel-trace:", r["Name"][:70], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
            if is_lockstep(r["Name"]):      # the lockstep instance (the persistent pre-roll launch is the RUN = true one)
                bench["roofline"]["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
                bench["roofline"]["rocprof_kernel_calls"] = int(r["Calls"])
pmc = {}
for grp in ("fetch", "write", "sq", "mfma", "mem"):
    f = one(f"pmc_{TAG}_{grp}/**/*counter_collection.csv")
    if not f:
        continue
    acc, cnt = {}, {}
    for r in csv.DictReader(open(f)):
        if not is_lockstep(r["Kernel_Name"]):
            continue        # lockstep launches only (warm-up + timed steps), not the persistent pre-roll launch
        k = r["Counter_Name"]
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
        cnt[k] = cnt.get(k, 0) + 1
    for k in acc:
        pmc[k] = acc[k] / cnt[k]
        pmc["launches_" + grp] = cnt[k]
pmc["note"] = "per-launch means over the lockstep step_kernel launches (warm-up + timed steps) of the SAME command as the bench line and the kernel trace (`bench.py --no-cpu-baseline --no-alt --no-configs --no-parity --steady 0` + the tag's arguments); one rocprofv3 --pmc pass per group"
with open(os.path.join(P, f"{TAG}_pmc.json"), "w") as f:
    json.dump(pmc, f, indent=1)
prec = bench["dtype"]
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fk, wk = pmc["FETCH_SIZE"], pmc["WRITE_SIZE"]   # KiB-ish units of 1 KB per the guide
    t = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), mean over the step-kernel launches; profiles/{TAG}_pmc.json",
         "fetch_kb_per_launch_raw": fk, "write_kb_per_launch": wk,
         "hbm_bytes_per_launch": (fk + wk) * 1024.0,
         "hbm_bytes_per_launch_fetch_x2_upper_bound": (2 * fk + wk) * 1024.0,
         "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
         "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-counts wide (16 B/lane) coalesced reads by 2x; this kernel reads 8 B/lane records, so the raw sum is reported as `traffic` and the x2-fetch figure as an upper bound.  With the stage records (AB'', gaps, cost gradients: 38.6 KB per instance in fp64) placed in global memory the traffic counted at the L2 boundary includes their write-out and the re-reads that miss L2: that is the price of running 4 instead of 2 instances per CU (DESIGN.md section 3.1)."}
    # identity of the build and of the command line: bench.py attaches the traffic only to a run of the same build and arguments
    sys.path.insert(0, ROOT)
    import bench as _bench
    cfg = bench["config"]
    t["source_sha16"] = _bench.kernel_source_sha16()
    t["args"] = {"steps": bench["steps"], "warmup": bench["warmup"], "preroll": cfg["preroll_periods"], "seed": int(os.environ.get("BENCH_SEED", "2026")),
                 "batch": cfg["batch_per_gpu"], "N": cfg["horizon_nodes"], "nb": cfg["rgp_basis"], "precision": prec}
    t["command"] = "python3 bench.py --no-cpu-baseline --no-alt --no-configs --no-parity --steady 0 " + open(os.path.join(O, f"args_{TAG}.txt")).read().strip()
    try:
        import subprocess
        t["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip() + " (+ working tree at collection time)"
    except Exception:
        t["commit"] = None
    with open(os.path.join(P, f"{ROUND}_pmc_traffic_{prec}_{TAG}.json"), "w") as f:
        json.dump(t, f, indent=1)
    print("traffic bytes/launch", t["hbm_bytes_per_launch"], "algorithmic", t["algorithmic_bytes_per_launch"])
    # the bench line of this collection was printed before these counters existed: record them in the tracked copy
    bench["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
    bench["roofline"]["traffic_source"] = t["source"]
    bench["roofline"]["traffic_note"] = t["note"]
    with open(os.path.join(P, f"{TAG}_bench.json"), "w") as f:
        json.dump(bench, f, indent=1)
print(json.dumps({k: v for k, v in pmc.items() if not k.startswith("launches") and k != "note"}, indent=1))
