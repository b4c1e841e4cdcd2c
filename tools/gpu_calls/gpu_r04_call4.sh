#!/bin/bash
# round 4, call 4: what the block order of a decomposed patch costs the solver's sweeps (tile alignment), before / after the
# halo set was made of whole 16 x 16-block patches; the N-rank tests at configs[3]'s per-rank size on the new order
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/gpu_halo_order_cost.py 2>&1 | tail -4
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_distributed.py -q -m gpu -p no:cacheprovider -s -k "configs3 or decomposed_step or cpp_mpi_driver_matches" > $OUT/r04c4_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "gpu_big|cup2d_run_mpi|passed|failed|FAILED|Error|assert" $OUT/r04c4_pytest.log | tail -30
