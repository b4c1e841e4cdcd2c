#!/bin/bash
# round 4, call 22: the paths that are not the default stay green: the three-launch full form as the process default; the generic
# N-rank exchange with three vectors travelling; the placement search off
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
CUP2D_FUSED_FORM=full timeout 1400 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider --deselect tests/test_solver_variants_gpu.py::test_forms_of_the_fused_sweeps \
  -k "not configs3_rank_size and not at_8192 and not in_place" > $OUT/r04_full_form_pytest.log 2>&1
echo "full form: pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04_full_form_pytest.log | tail -6
t0=$(date +%s)
CUP2D_COMM_DIRECT=0 CUP2D_GHOST_LOCAL=0 CUP2D_PLACEMENT_TRIES=0 timeout 1400 python3 -m pytest tests/test_distributed.py tests/test_comm.py tests/test_solver_variants_gpu.py -q -m gpu -p no:cacheprovider \
  -k "not in_place" > $OUT/r04_generic_pytest.log 2>&1
echo "generic exchange, no search: pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04_generic_pytest.log | tail -6
