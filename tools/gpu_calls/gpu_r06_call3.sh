#!/bin/bash
# round 6, call 3: where the folded pack differs (tools/gpu_fold_check.py), its timeline with the range compare instead of the
# per-tile flag; the rows of the general tiles applied inside the hybrid sweep behind a grid barrier (AMR tests, step time
# against the two-launch form); host stages of a regrid
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python3 tools/gpu_fold_check.py 2>&1 | grep -v "^W2026\|^E2026\|NCCL\|rccl\|RCCL" | tail -30
cd /tmp
for FOLD in 1 0; do
    rm -rf /tmp/prof_o
    CUP2D_FOLD_PACK=$FOLD NBX=512 NBY=512 AXES=xy STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r06c3_self_${FOLD}.log 2>&1
    echo "== fold $FOLD"; grep -E "ms/step|N-rank path" $GRAFT_REPO_ROOT/$OUT/r06c3_self_${FOLD}.log | cut -c1-160
    f=$(find /tmp/prof_o -name "*kernel_trace.csv" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r06c3_timeline_${FOLD}.txt | head -12
done
cd $GRAFT_REPO_ROOT
echo "== AMR tests (rows in the sweep)"
timeout 900 python3 -m pytest tests/test_amr.py -m gpu -x -q -p no:cacheprovider -k "fused_solver or poisson_solve or time_step or adapt_then_step or run_with_regridding" 2>&1 | tail -15
for ROWS in kernel launch; do
  echo "== AMR step, rows = $ROWS"
  CUP2D_HYB_ROWS=$ROWS LFINE=9 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step|sweep_|operator:" | cut -c1-200
done
echo "== regrid host stages"
CUP2D_HOST_TIMING=1 LFINE=9 timeout 400 python3 tools/gpu_amr_adapt_timing.py 2>&1 | grep -E "cup2d timing|adapt\(" | tail -40
