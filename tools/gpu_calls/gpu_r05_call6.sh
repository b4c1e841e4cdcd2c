#!/bin/bash
# round 5, call 6: the inner / halo phases of the adapted-grid block operators (one rank: public phases; 2 / 3 / 8 ranks: the
# default path with the ghost copies travelling while the inner blocks are swept, NaN-poisoned ghosts), per-test durations
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_amr.py tests/test_distributed.py tests/test_comm.py -q -m gpu -p no:cacheprovider --durations=15 \
  -k "two_phases or unsupported or kernels_bit_exact_gpu or amr_on_n_ranks or 4084 or one_rank_amr or cell_plan or time_step_vs_reference" > $OUT/r05c6_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s (call|setup)" $OUT/r05c6_pytest.log | tail -30
grep -n "first worker traceback" -A 25 $OUT/r05c6_pytest.log | cut -c1-300 | head -60
grep -n "error lines" -A 6 $OUT/r05c6_pytest.log | cut -c1-400 | head -30
grep -n "Error\|assert " $OUT/r05c6_pytest.log | cut -c1-300 | head -20
