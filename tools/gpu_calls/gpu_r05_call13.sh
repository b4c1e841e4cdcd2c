#!/bin/bash
# round 5, call 13 (record): the per-iteration timeline on the patch round 4 measured (its own W and E neighbour only), round 5's organisation
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_x
AXES=x STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_x -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r05c13_self_x.log 2>&1
grep -v "^W2026\|^E2026\|simple_timer" $GRAFT_REPO_ROOT/$OUT/r05c13_self_x.log | grep -E "ms/step|N-rank path" | cut -c1-200
f=$(find /tmp/prof_x -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r05_nrank_timeline_x.txt
