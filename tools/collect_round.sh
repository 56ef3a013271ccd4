#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_round.sh r6'): everything profiles/README.md lists for the final build of a round, in one call.
R=${1:-r6}
set -o pipefail   # a failing GPU suite fails the collection (advisor finding of round 5: the status of pytest was lost in the pipe)
fail=0
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "RCCL version|HIP version|ROCm version|Hostname|Librccl path" | tail -6 > $O/${R}_gpu_tests.log || fail=1
bash tools/collect_profiles.sh ${R}_k20 --gpus 1 --steps 20 --warmup 5 > $O/collect_${R}_k20.log 2>&1
python3 bench.py > $O/bench_${R}_default.json 2> $O/bench_${R}_default.err
bash tools/collect_swarm_traffic.sh $R > $O/collect_${R}_swarm.log 2>&1
LB_F32=1 bash tools/collect_swarm_traffic.sh $R > $O/collect_${R}_swarm_f32.log 2>&1
{ python tools/profile_phases.py f64 40; python tools/profile_phases.py f32 40; } > $O/${R}_phase_cycles.txt 2>&1
python tools/straggler_anatomy.py 200 > $O/${R}_straggler_anatomy.txt 2>&1
{ SOAK_EVERY=1 python tools/soak.py 1500 7; SOAK_EVERY=1 SOAK_NB=20 python tools/soak.py 600 7; SOAK_EVERY=1 SOAK_B=256 SOAK_N=50 SOAK_NB=50 python tools/soak.py 400 7; SOAK_B=8192 python tools/soak.py 600 7;
  SOAK_EVERY=1 python tools/soak.py 1000 2026; SOAK_EVERY=1 python tools/soak.py 1000 3;
  for s in 11 12 13 14 15 16; do SOAK_PREC=f64 SOAK_EVERY=1 python tools/soak.py 800 $s; done; SOAK_PREC=f64 SOAK_EVERY=1 SOAK_NB=20 python tools/soak.py 600 21; } > $O/${R}_soak.txt 2>&1
tools/microbench/chain_floor > $O/${R}_chain_floor.json 2>&1
# round 6: mpcq_tuning.groups on the four shapes; every f32 solve against the fp64 engine (five configurations)
bash tools/r6_groups.sh > $O/${R}_groups.txt 2>&1
{ python tools/f32_audit.py 1000 7; python tools/f32_audit.py 600 2026 300; python tools/f32_audit.py 800 3 100; SOAK_NB=20 python tools/f32_audit.py 500 11;
  SOAK_B=256 SOAK_N=50 SOAK_NB=50 python tools/f32_audit.py 300 7; python tools/f32_audit.py 300 2026; } > $O/${R}_f32_audit.txt 2>&1
cat $O/${R}_gpu_tests.log; tail -3 $O/${R}_soak.txt
[ $fail -eq 0 ] || { echo "GPU test suite FAILED (see $O/${R}_gpu_tests.log)"; exit 1; }
