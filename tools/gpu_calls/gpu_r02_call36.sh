#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
for lib in cup2d_amd/variants/libcup2d_hip_prev.so cup2d_amd/libcup2d_hip.so; do
  CUP2D_LIB=$PWD/$lib timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_all']
print('$lib', d['value'], d['ms_per_step_no_kernel_timers'], d['verified']['ok'], 'amr', d['amr_configs4']['value'], d['amr_configs4']['ms_per_step'])"
done; done
timeout 1200 python -m pytest tests/test_solver_variants_gpu.py tests/test_comm.py tests/test_distributed.py -m gpu -q 2>&1 | tail -3
