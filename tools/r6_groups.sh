#!/bin/bash
# Runs ON THE GPU BOX: mpcq_tuning.groups at the large-batch shapes and at the headline batch (lockstep periods of tools/large_batch.py)
O=gpurun_out; mkdir -p $O
LB_VARIANTS=g1,g2,g3,g4,g8 python3 tools/large_batch.py 8192 20 10 600 40 > $O/r6_groups_b8192.json 2> $O/r6_groups_b8192.err
LB_VARIANTS=g1,g2,g4 python3 tools/large_batch.py 8192 20 20 600 40 > $O/r6_groups_b8192_nb20.json 2> $O/r6_groups_b8192_nb20.err
LB_VARIANTS=g1,g2,g4,g8 python3 tools/large_batch.py 1024 20 10 600 200 > $O/r6_groups_b1024.json 2> $O/r6_groups_b1024.err
LB_VARIANTS=g1,g2,g4 python3 tools/large_batch.py 4096 50 50 300 20 > $O/r6_groups_b4096_n50.json 2> $O/r6_groups_b4096_n50.err
for f in $O/r6_groups_*.json; do python3 -c "
import json,sys
d=json.load(open('$f'))
print(d['B'],d['N'],d['nb'],[(r['order'],round(r['steps_per_s']/1e6,3)) for r in d['runs']],d['bitwise_equal'])"; done
tail -q -n 2 $O/r6_groups_*.err
