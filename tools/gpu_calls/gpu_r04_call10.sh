#!/bin/bash
# round 4, call 10: r' and p'' of the ghost blocks formed by the receiver, nu'' alone travels
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_comm.py tests/test_distributed.py -q -m gpu -p no:cacheprovider -s -k "in_place or configs3 or decomposed_step or cpp_mpi_driver_matches" 2>&1 | grep -E "gpu_big|passed|failed|FAILED|Error|assert" | tail -12
python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "plain|self-periodic \(|N-rank" | tail -4
CUP2D_GHOST_LOCAL=0 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "self-periodic \(|N-rank" | tail -3
NBX=512 NBY=256 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "plain|self-periodic \(|N-rank" | tail -4
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/$OUT/tl_self
STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_self -o tl -- python3 $R/tools/gpu_selfperiodic_step.py > $R/$OUT/tl_self.log 2>&1
cd $R
f=$(find $OUT/tl_self -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f "k_edge<3, 2" 40 | tee $OUT/r04_nrank_timeline_local.txt
