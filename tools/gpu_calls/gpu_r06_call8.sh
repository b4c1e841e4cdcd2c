#!/bin/bash
# round 6, call 8: reciprocals in pairs in the quad WENO5 walk (advect_walk.h rcp_pair): A/B against the build without them on one
# box (stage timers and floors inside whole steps), the FAST parity tests, the VMM placement probe
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
  timeout 200 python3 tools/gpu_advect_stages.py 4096 8 2>&1 | tail -1
  CUP2D_LIB=$PWD/tools/ab/libcup2d_hip_nopairs.so timeout 200 python3 tools/gpu_advect_stages.py 4096 8 2>&1 | tail -1
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 ./tools/placement_vmm.bin 6 2>&1 | tail -8
