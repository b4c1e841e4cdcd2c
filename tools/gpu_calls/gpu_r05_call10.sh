#!/bin/bash
# round 5, call 10: the two tests touched last (in-place test after its trim, bench at world 8 with four organisations)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_comm.py tests/test_bench_world8.py -q -m gpu -p no:cacheprovider --durations=5 -k "received_in_place or world_8" > $OUT/r05c10_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s call" $OUT/r05c10_pytest.log | tail -8
grep -n "Error\|assert \|Traceback" $OUT/r05c10_pytest.log | cut -c1-400 | head -20
