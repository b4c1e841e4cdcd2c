#!/bin/bash
# advect cache-policy variants inside the whole step (bench.py): v2 = temporal streams, nt1 = old values non-temporal, nt2 = results, default = both
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
for round in 1 2; do
  for lib in default v2 nt1 nt2; do
    if [ $lib = default ]; then unset CUP2D_LIB; else export CUP2D_LIB=$V/libcup2d_hip_walk_$lib.so; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-amr --no-verify 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$lib', 'step %.3f ms' % b['ms_per_step'], 'advect %.1f us' % (1e3*k['advect_stage']['ms_avg']), 'rhs %.1f' % (1e3*k['poisson_rhs']['ms_avg']), 'A %.1f C %.1f E %.1f' % (1e3*k['sweep_A']['ms_avg'],1e3*k['sweep_C']['ms_avg'],1e3*k['sweep_E']['ms_avg']))"
  done
done
