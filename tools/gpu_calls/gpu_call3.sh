#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_solver_variants_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_variants.log 2>&1; echo "pytest variants rc=$?"; tail -12 $OUT/pytest_variants.log
VARIANTS=${VARIANTS:-sweeps1,fused1} timeout 200 python tools/gpu_variants.py > $OUT/variants3.log 2>&1; echo "variants rc=$?"; grep -E "CHECK|TIME|VARIANTS|rror" $OUT/variants3.log | tail -12
for d in ${DBGS:-1 6 7}; do
  CUP2D_FUSED_DBG=$d SKIP_CHECK=1 VARIANTS=fused1 timeout 120 python tools/gpu_variants.py > $OUT/variants_dbg$d.log 2>&1
  echo "DBG=$d: $(grep -E '^TIME' $OUT/variants_dbg$d.log)"
done
