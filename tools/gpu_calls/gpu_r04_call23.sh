#!/bin/bash
# round 4, call 23: adapted grids on N ranks through the cell plans (only the cells the kernels read travel; ghost blocks start
# as NaN), the single-rank AMR path after amr_ghost3 moved into the shared header, and the placement assertion of the bench test
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_distributed.py tests/test_amr.py -q -s -m gpu -p no:cacheprovider -k "amr" > $OUT/r04c23_pytest.log 2>&1
echo "amr tests: pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|amr_big strips|amr_big:|cup2d_run_mpi -levelMax" $OUT/r04c23_pytest.log | tail -20
grep -E "Error|assert|Traceback" $OUT/r04c23_pytest.log | head -20
