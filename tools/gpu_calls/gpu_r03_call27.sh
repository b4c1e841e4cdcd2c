#!/bin/bash
# idle stream time between the launches of the two-launch organisation (kernel trace, no counters)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/gaps_r03
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_r03 -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers > $OUT/gaps_r03.log 2>&1; echo "rc=$?"
f=$(find $OUT/gaps_r03 -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_gaps.py $f | tee $OUT/r03_kernel_gaps.txt
rm -rf $OUT/gaps_r03
