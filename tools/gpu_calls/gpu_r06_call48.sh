#!/bin/bash
# round 6, call 48: the bench line with the sweeps sampled every 32nd iteration (16th before), the tests that read timers
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$?"; cat $OUT/final_bench.json | cut -c1-400
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
timeout 1200 python3 -m pytest tests/test_bench_world8.py tests/test_solver_variants_gpu.py tests/test_step_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
