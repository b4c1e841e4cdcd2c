#!/bin/bash
# the GPU suite with the three-launch full form as the process default (the path of custom preconditioners, the hybrid operator's
# relatives and CUP2D_FUSED_GHOST=edges must stay green next to the new default)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
CUP2D_FUSED_FORM=full timeout 1400 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider --deselect tests/test_solver_variants_gpu.py::test_forms_of_the_fused_sweeps > $OUT/full_form_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/full_form_pytest.log | tail -10
