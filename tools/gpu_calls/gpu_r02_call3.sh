#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_distributed.py tests/test_baseline_sizes_gpu.py -m gpu -q -s -k "not functor" --durations=5 > $OUT/r02_pytest3.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r02_pytest3.log
for v in "" cup2d_amd/variants/libcup2d_hip_base.so; do
  echo "== variant ${v:-new}"
  CUP2D_LIB=${v:+$PWD/$v} VARIANTS=fused1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "CHECK|TIME|VARIANTS|Error|error" | tail -6
done 2>&1 | tee $OUT/r02_variants3.log
echo "total $(( $(date +%s) - t0 )) s"
