#!/bin/bash
# round 4, call 6: the timeline of one iteration on the N-rank path (self-periodic patch, bytes through RCCL)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rm -rf $GRAFT_REPO_ROOT/$OUT/tl_self
STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tl_self -o tl -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/tl_self.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/tl_self -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f "k_edge<3, 2" 40 | tee $OUT/r04_nrank_timeline.txt
