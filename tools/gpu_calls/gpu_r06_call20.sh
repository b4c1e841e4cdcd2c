#!/bin/bash
# round 6, call 20: the tables as before (every block of a general tile is k_hyb_rows'), the two-launch organisation on request:
# the AMR / assembled-operator tests, then both forms on the 63 k-block grid
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_amr.py tests/test_spmat_gpu.py -x -q -m gpu -p no:cacheprovider -k "tile_fused_solver or installed_from_the_tables or spmat or hybrid or matrix" -s > $OUT/c20_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "two launches|passed|failed|^FAILED|^ERROR|Error|assert " $OUT/c20_pytest.log | cut -c1-250 | tail -12
for F in auto eab auto eab; do
  FORM=$F LFINE=9 NOTIMING=1 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|Traceback|Error" | cut -c1-200
done
