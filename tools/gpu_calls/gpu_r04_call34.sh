#!/bin/bash
# round 4, call 34: the uniform 2048^2 and 4096^2 steps as the GPU sees them after the step lost its two host waits
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/$OUT/tl_2048b $R/$OUT/tl_4096b
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_2048b -o tl -- python3 $R/bench.py --n 2048 --steps 4 --warmup 2 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers --no-nrank-proxy > $R/$OUT/tl_2048b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl_4096b -o tl -- python3 $R/bench.py --n 4096 --steps 4 --warmup 2 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers --no-nrank-proxy > $R/$OUT/tl_4096b.log 2>&1
cd $R
python3 tools/kernel_step_timeline.py $(find $OUT/tl_2048b -name "*kernel_trace.csv" | head -1) "k_pressure_rhs" -2 | tee $OUT/r04b_2048_step_timeline.txt
python3 tools/kernel_step_timeline.py $(find $OUT/tl_4096b -name "*kernel_trace.csv" | head -1) "k_pressure_rhs" -2 | tee $OUT/r04b_4096_step_timeline.txt
find $OUT/tl_2048b $OUT/tl_4096b -name "*kernel_trace.csv" -delete
