#!/bin/bash
# Runs ON THE GPU BOX: round-6 baseline of the large-batch shape (configs[3] per rank) on the build at hand -- launch times, then the
# counter passes the round-5 verdict asked for (instruction cache, issue utilisation) over the lockstep launches of the compact layout.
TAG=${1:-r6base}
O=gpurun_out; mkdir -p $O
LB_VARIANTS=compact python3 tools/large_batch.py 8192 20 10 600 40 > $O/${TAG}_lb.json 2> $O/${TAG}_lb.err
LB_VARIANTS=compact python3 tools/large_batch.py 8192 20 20 600 40 > $O/${TAG}_lb_nb20.json 2> $O/${TAG}_lb_nb20.err
LB_PREROLL_LOCKSTEP=1 bash tools/pmc_large_batch.sh $TAG "8192 20 10 200 20" compact > $O/${TAG}_pmc.log 2>&1
cat $O/${TAG}_lb.json $O/${TAG}_lb_nb20.json; tail -3 $O/${TAG}_pmc.log
