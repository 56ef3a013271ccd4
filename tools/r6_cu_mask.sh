#!/bin/bash
# Runs ON THE GPU BOX: does the large-batch rate scale with the number of CUs?  The same lockstep launches (compact layout, one and two
# groups) on all 256 CUs and on a half / a quarter of them (HSA_CU_MASK of the ROCm runtime; the engine still sees 256 CUs, so layout
# and groups are forced and the batch is scaled with the CUs: same number of rounds of workgroups per launch).  A rate per CU that
# RISES with fewer CUs means a resource shared beyond the CU (L2 / fabric / HBM latency under load) bounds the full device.
O=gpurun_out; mkdir -p $O
run() {  # tag mask B
  if [ "$2" = none ]; then
    LB_VARIANTS=compact,compact_g2 timeout 600 python3 tools/large_batch.py $3 20 10 600 40 > $O/r6_cumask_$1.json 2> $O/r6_cumask_$1.err
  else
    HSA_CU_MASK="$2" LB_VARIANTS=compact,compact_g2 timeout 600 python3 tools/large_batch.py $3 20 10 600 40 > $O/r6_cumask_$1.json 2> $O/r6_cumask_$1.err
  fi
  python3 -c "
import json
d=json.load(open('$O/r6_cumask_$1.json'))
print('$1', 'mask=$2', 'B', d['B'], [(r['order'][:40], round(r['steps_per_s']/1e6,3), round(r['kernel_avg_ms'],4)) for r in d['runs']])" || tail -3 $O/r6_cumask_$1.err
}
run full none 8192
run half 0:0-127 4096
run half_b8192 0:0-127 8192
run quarter 0:0-63 2048
run eighth 0:0-31 1024
run full_b4096 none 4096
