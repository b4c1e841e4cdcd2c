#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_u -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-amr --no-kernel-timers --no-verify > /dev/null 2>&1
echo "uniform 4096^2, no kernel timers:"; python $R/tools/kernel_gaps.py $(find /tmp/gap_u -name "*kernel_trace.csv" | head -1)
cd /tmp && NOTIMING=1 LFINE=9 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_a -o t -- python $R/tools/gpu_amr_bench.py > /dev/null 2>&1
echo "AMR 63k blocks, no per-launch events:"; python $R/tools/kernel_gaps.py $(find /tmp/gap_a -name "*kernel_trace.csv" | head -1)
