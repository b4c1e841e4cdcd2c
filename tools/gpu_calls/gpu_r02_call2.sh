#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_spmat_gpu.py tests/test_baseline_sizes_gpu.py tests/test_comm.py -m gpu -q -s -k "b2 or bench_configuration or comm" --durations=5 > $OUT/r02_pytest2.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/r02_pytest2.log
bash tools/gpu_r02_sq.sh r02
echo "total $(( $(date +%s) - t0 )) s"
