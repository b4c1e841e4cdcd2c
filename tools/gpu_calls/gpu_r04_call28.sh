#!/bin/bash
# round 4, call 28: stored rows of the hybrid operator with the next group's columns requested ahead, stored slices first in the
# rows launch's list: the AMR step and the tests of everything that applies stored rows (k_sell, k_hyb_rows)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
LFINE=9 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|sweep|scalars|operator:" | head -8
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_amr.py tests/test_spmat_gpu.py tests/test_comm.py -q -m gpu -p no:cacheprovider > $OUT/r04c28_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c28_pytest.log | tail -8
