#!/bin/bash
# round 4, call 24: the entry points that said "unsupported" on adapted grids (one RK stage; a solve with no installed operator)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_amr.py -q -m gpu -p no:cacheprovider > $OUT/r04c24_pytest.log 2>&1
echo "test_amr: pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c24_pytest.log | tail -8
grep -E "Error|assert " $OUT/r04c24_pytest.log | head -20
