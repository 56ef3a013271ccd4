#!/bin/bash
# Runs ON THE GPU BOX: what a tighter settle threshold of the mixed-precision refinement would give and cost (experiment build
# `make variant NAME=vtc EXTRA=-DMPCQ_MIXED_TOLC=2.5e-7`; the product settles at 1e-6).  Audits of the configurations that held the worst
# status-0 deviations, and the f32 lockstep rates, product and variant.
O=gpurun_out; mkdir -p $O
for lib in mpc_quad_ros_amd/libmpcq.so mpc_quad_ros_amd/libmpcq_vtc.so; do
  n=$(basename $lib .so)
  echo "==== $n"
  MPCQ_LIB=$lib SOAK_B=8192 python3 tools/f32_audit.py 100 4 300 2>/dev/null | head -5
  MPCQ_LIB=$lib python3 tools/f32_audit.py 600 2026 300 2>/dev/null | head -5
  MPCQ_LIB=$lib SOAK_B=512 SOAK_N=50 SOAK_NB=50 python3 tools/f32_audit.py 300 9 0 2>/dev/null | head -5
  MPCQ_LIB=$lib LB_F32=1 LB_VARIANTS=g2 python3 tools/large_batch.py 8192 20 10 600 60 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin)['runs'][0]; print('f32 B=8192 g2', round(r['steps_per_s']/1e6,3), 'M')"
  MPCQ_LIB=$lib LB_F32=1 LB_VARIANTS=g1 python3 tools/large_batch.py 1024 20 10 600 200 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin)['runs'][0]; print('f32 B=1024', round(r['steps_per_s']/1e6,3), 'M')"
done
