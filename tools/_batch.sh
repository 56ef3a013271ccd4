python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "compact or variants_agree or block_order or config2_full or config3 or config4_full or long_horizon" 2>&1 | tail -4
for c in "8192 20 10 600 40" "8192 20 20 300 40" "4096 50 50 300 20" "2048 20 10 600 40" "4096 20 10 600 40"; do LB_VARIANTS=global,compact python tools/large_batch.py $c > gpurun_out/r4_lb_tmp.json 2>gpurun_out/r4_lb_tmp.err; tail -c 300 gpurun_out/r4_lb_tmp.err; python -c "
import json
d=json.load(open('gpurun_out/r4_lb_tmp.json'))
print(d['B'],d['N'],d['nb'],[(r['order'][8:24],round(r['steps_per_s']/1e6,3),round(r['kernel_avg_ms'],4)) for r in d['runs']])
"; done
