#!/bin/bash
# round 5, call 7: the world-8 tests after their solves were shortened (durations), the halo_plan refusal under a live communicator
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python3 -m pytest tests/test_comm.py tests/test_distributed.py -q -m gpu -p no:cacheprovider --durations=8 \
  -k "8-2-4 or two_by_four or fills_the_ghost" > $OUT/r05c7_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s call" $OUT/r05c7_pytest.log | tail -14
grep -n "first worker traceback" -A 25 $OUT/r05c7_pytest.log | cut -c1-300 | head -50
grep -n "error lines" -A 6 $OUT/r05c7_pytest.log | cut -c1-400 | head -20
grep -n "Error\|assert " $OUT/r05c7_pytest.log | cut -c1-300 | head -12
