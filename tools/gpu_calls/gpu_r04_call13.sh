#!/bin/bash
# round 4, call 13: where a block-AMR step's GPU time and idle time go (63 k-block grid), and the same for the uniform 2048^2 step
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/$OUT/tl_amr $R/$OUT/tl_2048
LFINE=9 NOTIMING=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_amr -o tl -- python3 $R/tools/gpu_amr_bench.py > $R/$OUT/tl_amr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_2048 -o tl -- python3 $R/bench.py --n 2048 --steps 4 --warmup 2 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers --no-nrank-proxy > $R/$OUT/tl_2048.log 2>&1
cd $R
grep "AMR step" $OUT/tl_amr.log
python3 tools/kernel_step_timeline.py $(find $OUT/tl_amr -name "*kernel_trace.csv" | head -1) "k_amr_vector<1>" -2 | tee $OUT/r04_amr_step_timeline.txt
python3 tools/kernel_step_timeline.py $(find $OUT/tl_2048 -name "*kernel_trace.csv" | head -1) "k_pressure_rhs" -2 | tee $OUT/r04_2048_step_timeline.txt
