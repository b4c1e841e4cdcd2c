#!/bin/bash
# quad advect kernel: non-temporal streams (nt1 = old values of stage 2, nt2 = results, nt3 = both), per stage, interleaved
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
for round in 1 2 3; do
  for lib in default nt1 nt2 nt3; do
    if [ $lib = default ]; then unset CUP2D_LIB; else export CUP2D_LIB=$V/libcup2d_hip_walk_$lib.so; fi
    timeout 300 python tools/gpu_advect_stages.py 4096 20 2>&1 | tail -1 | sed "s/^/$lib /" | sed 's/us per launch (host clock over 20 back-to-back launches)//'
  done
done
