#!/bin/bash
# round 6, call 33: the GPU suite, smoke and the bench line on the placement search with the repair
# the placement search that goes on until a fast and a slow set have shown
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > $OUT/c33_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|s call" $OUT/c33_pytest.log | tail -12
grep -n "Error\|assert " $OUT/c33_pytest.log | cut -c1-300 | head -10
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c33_bench.json 2> $OUT/c33_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; wc -c $OUT/c33_bench.json; cat $OUT/c33_bench.json
cp $OUT/bench_detail.json $OUT/c33_bench_detail.json
grep -E "tune_placement: (set|search)" $OUT/c33_bench.err | cut -c1-160 | head -60
