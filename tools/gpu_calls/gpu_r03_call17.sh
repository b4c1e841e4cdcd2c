#!/bin/bash
# the GPU suite without -x (development: every failure of a change at once)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1400 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/dev_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/dev_pytest.log | tail -30
