#!/bin/bash
# round 6, call 55: after the last change to tune_placement: the solver tests at the BASELINE sizes, the forms, the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_solver_variants_gpu.py tests/test_baseline_sizes_gpu.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/final_bench.json')); s=d['summary']
print(d['value'], d['ms_per_step'], s.get('ms_per_step_no_kernel_timers'), s['second_size_2048']['value'], s['amr_configs4']['value'], s['placement'], d['verified_ok'])"
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
grep -E "repair" $OUT/final_bench.err | head -4 | cut -c1-250
