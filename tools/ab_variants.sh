#!/bin/bash
# Runs ON THE GPU BOX: A/B of experiment builds / environment switches on the bench workload (lockstep kernel time and
# pass histogram).  usage: bash tools/ab_variants.sh OUTDIR "label|ENV=.. ENV=.." ...
O=$1; shift
mkdir -p $O
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --no-cpu-baseline --no-alt > $O/bench_$label.json 2> $O/bench_$label.err
  env $envs python tools/pass_hist.py f64 200 > $O/hist_$label.txt 2>&1
  python - "$O/bench_$label.json" "$label" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:12s} lockstep {d['value']/1e6:6.3f} M/s  kernel avg {r['kernel_avg_ms']:.4f} min {r['kernel_min_ms']:.4f} max {r['kernel_max_ms']:.4f}  passes mean {d['solver']['mean_qp_passes']:.3f} max {d['solver']['max_qp_passes']}")
PY
done
