#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amr.py tests/test_spmat_gpu.py tests/test_solver_variants_gpu.py -m gpu -q > $OUT/r02_pytest19.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r02_pytest19.log
for lf in 8 9; do echo "== LFINE $lf merged"; LFINE=$lf timeout 300 python tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|scalars|sweep_B"; echo "== LFINE $lf finish launches"; CUP2D_FINISH_IN_KERNEL=0 LFINE=$lf timeout 300 python tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|scalars|sweep_B"; done
