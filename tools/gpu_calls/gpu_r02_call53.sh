#!/bin/bash
# quad advect kernel with non-temporal old / result streams: event-timed launches, parity at 4096^2, L2-miss traffic
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/gpu_advect_only.py 4096 10 check 2>&1 | tail -2
CUP2D_LIB=$PWD/cup2d_amd/variants/libcup2d_hip_walk_v2.so timeout 300 python tools/gpu_advect_only.py 4096 10 2>&1 | tail -1 | sed 's/^/v2: /'
bash tools/gpu_quick_traffic.sh walk python tools/gpu_advect_only.py 4096 2 2>&1 | grep "advect\|rc="
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advect or rectangular or step_matches or functors_vs_golden or consecutive" 2>&1 | tail -2
