#!/bin/bash
# quad advect kernel: interleaved A/B of the library versions (same box, three rounds)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/cup2d_amd/variants
for round in 1 2 3; do
  for lib in default v3p v2 v1; do
    if [ $lib = default ]; then unset CUP2D_LIB; else export CUP2D_LIB=$V/libcup2d_hip_walk_$lib.so; fi
    for pr in 1 2; do
      [ $pr = 2 ] && [ $lib != v3p ] && continue
      CUP2D_WALK_PRIO=$pr timeout 300 python tools/gpu_advect_only.py 4096 10 2>&1 | tail -1 | sed "s/^/$lib prio=$pr /" | awk '{for(i=1;i<=NF;i++) if ($i=="us") printf "%s %s %s us\n",$1,$2,$(i-1)}'
    done
  done
done
