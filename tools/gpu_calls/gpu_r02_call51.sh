#!/bin/bash
# the GPU tests the quad advect kernel can touch: parity file, N ranks on one GPU (ghost blocks, inner / halo phases), seam B2
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_comm.py tests/test_spmat_gpu.py -m gpu -q --durations=6 2>&1 | tail -14
