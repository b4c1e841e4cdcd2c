#!/bin/bash
# round 6, call 50: the bench line with 20 setup steps in front of the warm-up (and the sweeps sampled every 32nd iteration); the world-8 bench test
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$? $(wc -c < $OUT/final_bench.json) bytes"; python3 -c "
import json; d=json.load(open('$OUT/final_bench.json')); s=d['summary']
print(d['value'], d['ms_per_step'], s.get('ms_per_step_no_kernel_timers'), d['config'].get('setup_steps'), s['second_size_2048']['value'], s['amr_configs4']['value'], s['placement'], d['verified_ok'])"
done
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
timeout 1200 python3 -m pytest tests/test_bench_world8.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
