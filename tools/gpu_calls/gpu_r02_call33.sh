#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
for lib in cup2d_amd/variants/libcup2d_hip_prev.so cup2d_amd/libcup2d_hip.so; do
  CUP2D_LIB=$PWD/$lib timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-amr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_all']
print('$lib', d['value'], d['ms_per_step_no_kernel_timers'], {k:r[k]['avg_launch_ms'] for k in ('sweep_A','sweep_C','sweep_E')})"
done; done
