#!/bin/bash
# round 4, call 30: the GPU suite once more with the driver's command line on another box (flakiness check)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/r04c30_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/r04c30_pytest.log
rocm-smi --showuniqueid 2>&1 | grep -E "Unique ID:" | head -1
