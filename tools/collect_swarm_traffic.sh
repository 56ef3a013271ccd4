#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic and kernel time of the large-batch lockstep launches (BASELINE configs[3] per rank: B = 8192,
# N = 20, nb = 10, compact layout, cost-sorted launch order) -> gpurun_out/<TAG>_pmc_traffic_swarm_b8192.json + kernel stats.
# Three runs of the same program (kernel trace; FETCH_SIZE; WRITE_SIZE), the program itself behind `--`.
TAG=${1:-r4}      # LB_F32=1 in the environment: the f32 (mixed-precision) engine; the output files get the suffix _f32
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export LB_VARIANTS=compact LB_PREROLL_LOCKSTEP=1
ARGS="8192 20 10 300 40"
SUF=""; [ -n "$LB_F32" ] && SUF="_f32" && TAG="${TAG}f32"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_swarm -- python3 tools/large_batch.py $ARGS > $O/prof_${TAG}_swarm.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_swarm_fetch -- python3 tools/large_batch.py $ARGS > $O/pmc_${TAG}_swarm_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_swarm_write -- python3 tools/large_batch.py $ARGS > $O/pmc_${TAG}_swarm_write.log 2>&1
python3 - "$TAG" "$SUF" <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
tag, suf = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
def mean_counter(sub, name):
    acc = n = 0
    for f in glob.glob(f"gpurun_out/pmc_{tag}_swarm_{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                acc += float(r["Counter_Value"]); n += 1
    return acc / max(n, 1), n
fetch, nf = mean_counter("fetch", "FETCH_SIZE")
write, nw = mean_counter("write", "WRITE_SIZE")
kavg = oavg = None
for f in glob.glob(f"gpurun_out/prof_{tag}_swarm/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Name"]:
            kavg = float(r["AverageNs"]) / 1e3
        if "order_kernel" in r["Name"]:
            oavg = float(r["AverageNs"]) / 1e3
    os.system(f"cp {f} gpurun_out/{tag}_swarm_kernel_stats.csv")
alg = 8192 * bench.algorithmic_bytes(20, 10, 4 if suf else 8)
out = {"workload": "BASELINE configs[3] per rank: B = 8192, N = 20, nb = 10, compact layout, cost-sorted launch order; tools/large_batch.py 8192 20 10 300 40 "
                   "with LB_VARIANTS=compact LB_PREROLL_LOCKSTEP=1 (345 lockstep launches: 300 from hover + 5 + 40)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), mean over the step-kernel launches; kernel average from "
                 "rocprofv3 --kernel-trace --stats of the same command (tools/collect_swarm_traffic.sh)",
       "launches_counted": [nf, nw], "fetch_kb_per_launch_raw": fetch, "write_kb_per_launch": write,
       "hbm_bytes_per_launch": (fetch + write) * 1024, "hbm_bytes_per_launch_fetch_x2_upper_bound": (2 * fetch + write) * 1024,
       "algorithmic_bytes_per_launch": alg, "kernel_avg_us_over_these_launches": kavg, "order_kernel_avg_us": oavg,
       "source_sha16": bench.kernel_source_sha16()}
out["precision"] = "f32 (mixed precision)" if suf else "f64"
json.dump(out, open(f"gpurun_out/{tag[:-3] if suf else tag}_pmc_traffic_swarm_b8192{suf}.json", "w"), indent=1)
print(json.dumps(out))
PY
