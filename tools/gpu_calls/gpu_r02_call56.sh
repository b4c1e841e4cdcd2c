#!/bin/bash
# advect: 32-bit offsets on scalar bases for every stream + sign flags in scalar registers (new) against v4, inside the whole step; parity
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advect or rectangular or step_matches or functors_vs_golden or consecutive" 2>&1 | tail -2
for round in 1 2 3; do
  for lib in default v5; do
    if [ $lib = default ]; then unset CUP2D_LIB; else export CUP2D_LIB=$V/libcup2d_hip_walk_$lib.so; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-amr --no-verify 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$lib', 'step %.3f ms' % b['ms_per_step'], 'advect %.1f us' % (1e3*k['advect_stage']['ms_avg']), 'rhs %.1f' % (1e3*k['poisson_rhs']['ms_avg']))"
  done
done
timeout 300 python tools/gpu_advect_only.py 4096 3 check 2>&1 | tail -1
