#!/bin/bash
# round 6, call 17: the two-launch organisation on the hybrid operator (k_edge HYB + k_hyb_rows MODE 2 / 3): parity, then A/B
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_amr.py -x -q -m gpu -p no:cacheprovider -k "tile_fused_solver or installed_from_the_tables" -s > $OUT/c17_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "two launches|passed|failed|^FAILED|^ERROR|Error|assert " $OUT/c17_pytest.log | cut -c1-250 | tail -30
for F in full auto full auto; do
  FORM=$F LFINE=9 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|sweep_|Traceback|Error" | cut -c1-200
done
