#!/bin/bash
# Runs ON THE GPU BOX: counter passes over the lockstep launches of a large batch (tools/large_batch.py), one layout per run.
#   usage: bash tools/pmc_large_batch.sh TAG "B N nb preroll steps" VARIANT
TAG=$1; ARGS=$2; VAR=$3
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export LB_VARIANTS=$VAR
[ -f $O/counters_list.txt ] || rocprofv3 -L > $O/counters_list.txt 2>&1
run() {  # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${TAG}_$n -- python3 tools/large_batch.py $ARGS > $O/pmc_${TAG}_$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run mem TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
python3 - "$TAG" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
    acc, cnt = {}, {}
    for r in csv.DictReader(open(f)):
        if "step_kernel" not in r["Kernel_Name"] or ", true>" in r["Kernel_Name"].split("Cfg<")[1][:40].replace(", true, true>", ", XX>") and False:
            continue
        run = "true" in r["Kernel_Name"].split("Cfg<")[1].split(">")[0].split(",")[4]
        if run:
            continue   # the persistent pre-roll launch
        k = r["Counter_Name"]
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
    for k in acc:
        out[k] = acc[k] / cnt[k]; out["launches"] = cnt[k]
json.dump(out, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps(out))
PY
grep -i -c "icache\|ifetch" $O/counters_list.txt
