#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh TAG [bench args]'): the bench line, the rocprofv3 kernel
# trace summary and the PMC passes (each in its own run, kernel-trace only) the numbers in profiles/ come from.  Every
# pass runs THE SAME bench command (same pre-roll, warm-up and steps), so kernel time, counters and traffic describe
# the same launches; the program itself follows `--` (no env / bash -c hop under the profiler).
TAG=${1:-r4_final_k20}; shift
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--no-cpu-baseline --no-alt --no-configs --no-parity --steady 0 $@"   # the profiled program: headline leg only (same pre-roll, warm-up, steps; no steady-state / latency legs)
echo "$@" > $O/args_$TAG.txt
python3 bench.py $@ > $O/bench_$TAG.json 2> $O/bench_$TAG.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 bench.py $ARGS > $O/prof_$TAG.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_fetch -- python3 bench.py $ARGS > $O/pmc_${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_write -- python3 bench.py $ARGS > $O/pmc_${TAG}_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${TAG}_sq -- python3 bench.py $ARGS > $O/pmc_${TAG}_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_mfma -- python3 bench.py $ARGS > $O/pmc_${TAG}_mfma.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${TAG}_mem -- python3 bench.py $ARGS > $O/pmc_${TAG}_mem.log 2>&1
tail -1 $O/bench_$TAG.json | cut -c1-400
