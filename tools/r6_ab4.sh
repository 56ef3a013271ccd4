#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds on the four BASELINE shapes (lockstep periods of tools/large_batch.py, automatic groups).
# usage: bash tools/r6_ab4.sh TAG lib1.so lib2.so ...
TAG=$1; shift
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  MPCQ_LIB=$lib LB_VARIANTS=g2 python3 tools/large_batch.py 8192 20 10 600 40 > $O/ab4_${TAG}_${n}_a.json 2> $O/ab4_${TAG}_${n}.err
  MPCQ_LIB=$lib LB_VARIANTS=g1 python3 tools/large_batch.py 1024 20 10 600 200 > $O/ab4_${TAG}_${n}_b.json 2>> $O/ab4_${TAG}_${n}.err
  MPCQ_LIB=$lib LB_VARIANTS=g2 python3 tools/large_batch.py 8192 20 20 300 30 > $O/ab4_${TAG}_${n}_c.json 2>> $O/ab4_${TAG}_${n}.err
  MPCQ_LIB=$lib LB_VARIANTS=g2 python3 tools/large_batch.py 4096 50 50 300 20 > $O/ab4_${TAG}_${n}_d.json 2>> $O/ab4_${TAG}_${n}.err
  python3 - $O/ab4_${TAG}_${n} $n <<'PY'
import json, sys
r = [json.load(open(f"{sys.argv[1]}_{k}.json"))["runs"][0] for k in "abcd"]
print(f"{sys.argv[2]:18s} B=8192/nb10 x2 {r[0]['steps_per_s']/1e6:7.3f} M | B=1024 {r[1]['steps_per_s']/1e6:6.3f} M | B=8192/nb20 x2 {r[2]['steps_per_s']/1e6:7.3f} M | B=4096/N50 x2 {r[3]['steps_per_s']/1e6:6.3f} M   digests {r[0]['digest']:.4f} {r[1]['digest']:.4f}")
PY
done
done
