#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amr.py -m gpu -q -x -s -k "tile_fused" > $OUT/r02_pytest24a.log 2>&1; echo "fused hybrid rc=$?"; grep -E "blocks:|passed|failed|Error|assert" $OUT/r02_pytest24a.log | head -20
timeout 1500 python -m pytest tests/test_amr.py tests/test_spmat_gpu.py -m gpu -q > $OUT/r02_pytest24b.log 2>&1; echo "amr+spmat rc=$?"; tail -5 $OUT/r02_pytest24b.log
LFINE=9 timeout 600 python tools/gpu_amr_bench.py > $OUT/r02_amr24.log 2>&1; tail -14 $OUT/r02_amr24.log
