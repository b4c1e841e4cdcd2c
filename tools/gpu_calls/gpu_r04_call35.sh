#!/bin/bash
# round 4, call 35: the N-rank adapted-grid tests and the RCCL-to-self cell plan after the empty-plan semantics of
# cup2d_halo_plan_cells changed (0, 0 = a plan; -1, -1 = remove)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 700 python3 -m pytest tests/test_distributed.py tests/test_comm.py -q -m gpu -p no:cacheprovider -k "amr or cell_plan" > $OUT/r04c35_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c35_pytest.log | tail -5
