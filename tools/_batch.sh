python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_c_bench_k20.json 2> gpurun_out/r4_c_bench_k20.err; tail -c 300 gpurun_out/r4_c_bench_k20.err
python bench.py > gpurun_out/r4_c_bench_default.json 2> gpurun_out/r4_c_bench_default.err; tail -c 300 gpurun_out/r4_c_bench_default.err
python tools/ablation_config5.py > gpurun_out/r4_ablation.log 2>&1; tail -6 gpurun_out/r4_ablation.log | cut -c1-300
