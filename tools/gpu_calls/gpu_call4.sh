#!/bin/bash
# smoother tile kernels: parity tests, AMR kernels after the amr_ghost.h refactor, staged multi-rank, timing
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "jacobi" > gpurun_out/pytest_smoother.log 2>&1; echo "pytest smoother rc=$?"; tail -5 gpurun_out/pytest_smoother.log
timeout 600 python tools/gpu_smoother.py > gpurun_out/smoother_timing.log 2>&1; echo "timing rc=$?"; cat gpurun_out/smoother_timing.log
timeout 900 python -m pytest tests/test_amr.py tests/test_distributed.py -m gpu -q -x > gpurun_out/pytest_amr_dist.log 2>&1; echo "pytest amr+dist rc=$?"; tail -5 gpurun_out/pytest_amr_dist.log
