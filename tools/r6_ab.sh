#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds on the lockstep periods of tools/large_batch.py.  usage: bash tools/r6_ab.sh TAG lib1.so lib2.so ...
TAG=$1; shift
O=gpurun_out; mkdir -p $O
for lib in "$@"; do
  n=$(basename $lib .so)
  MPCQ_LIB=$lib LB_VARIANTS=g2 python3 tools/large_batch.py 8192 20 10 600 40 > $O/ab_${TAG}_${n}_b8192.json 2> $O/ab_${TAG}_${n}.err
  MPCQ_LIB=$lib LB_VARIANTS=g1 python3 tools/large_batch.py 1024 20 10 600 200 > $O/ab_${TAG}_${n}_b1024.json 2>> $O/ab_${TAG}_${n}.err
  python3 - $O/ab_${TAG}_${n}_b8192.json $O/ab_${TAG}_${n}_b1024.json $n <<'PY'
import json, sys
a, b = (json.load(open(f))["runs"][0] for f in sys.argv[1:3])
print(f"{sys.argv[3]:22s} B=8192 (2 groups) {a['steps_per_s']/1e6:7.3f} M  digest {a['digest']:.6f} | B=1024 {b['steps_per_s']/1e6:6.3f} M  kernel {b['kernel_avg_ms']:.4f} ms  digest {b['digest']:.6f}")
PY
done
