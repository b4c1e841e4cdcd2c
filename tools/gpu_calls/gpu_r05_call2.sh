#!/bin/bash
# round 5, call 2: the three world-8 failures of call 1 with the workers' own tracebacks
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_bench_world8.py tests/test_distributed.py -q -m gpu -p no:cacheprovider -s \
  -k "world_8 or 8-2-4 or two_by_four" > $OUT/r05c2_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|gpu_big" $OUT/r05c2_pytest.log | tail -12
grep -n "first worker traceback" -A 45 $OUT/r05c2_pytest.log | head -150
