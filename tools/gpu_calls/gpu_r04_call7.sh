#!/bin/bash
# round 4, call 7: ghost blocks received in place on the compute stream: the test, the step time and the timeline of an iteration
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_comm.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -v "^\[" | grep -E "plain|self-periodic|N-rank|sweep" | tail -6
CUP2D_COMM_DIRECT=0 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "self-periodic \(|N-rank" | tail -3
NBX=512 NBY=256 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "plain|self-periodic \(|N-rank" | tail -4
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/$OUT/tl_self
STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_self -o tl -- python3 $R/tools/gpu_selfperiodic_step.py > $R/$OUT/tl_self.log 2>&1
cd $R
f=$(find $OUT/tl_self -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f "k_edge<3, 2" 40 | tee $OUT/r04_nrank_timeline_direct.txt
