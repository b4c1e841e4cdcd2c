#!/bin/bash
# round 6, the last commit as the driver runs it: the GPU suite, smoke, the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/final_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/final_pytest.log | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; wc -c $OUT/final_bench.json; cat $OUT/final_bench.json
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
# the step as the GPU sees it
cd /tmp; rm -rf /tmp/prof_s
CUP2D_BENCH_DETAIL=/tmp/d.json timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_step_timeline.py $f "k_pressure_rhs" 3 2>&1 | head -24 | tee $GRAFT_REPO_ROOT/$OUT/r06_4096_step_timeline.txt
