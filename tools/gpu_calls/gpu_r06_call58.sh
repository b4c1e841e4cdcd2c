#!/bin/bash
# round 6, call 58: the bench line on the round's last commit (the search's criterion is the median now)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$? $(wc -c < $OUT/final_bench.json) bytes"; python3 -c "
import json; d=json.load(open('$OUT/final_bench.json')); s=d['summary']
print(d['value'], d['ms_per_step'], s.get('ms_per_step_no_kernel_timers'), s['second_size_2048']['value'], s['amr_configs4']['value'], s['amr_configs4']['regrid_ms'], s['placement'], d['verified_ok'])
print({k:(v['ratio_to_plain'], v['value']) for k,v in s['nrank_path_on_one_gpu'].items() if 'paper' not in k}, s['nrank_path_on_one_gpu']['configs3_on_paper'])"
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
grep -E "search" $OUT/final_bench.err | cut -c1-100
