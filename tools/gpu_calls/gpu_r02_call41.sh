#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for n in 2048 8192; do
timeout 900 python bench.py --n $n --steps 6 --warmup 2 --no-cpu-baseline --no-amr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_all']
print($n, d['value'], d['ms_per_step_no_kernel_timers'], d['verified']['ok'], {k:(r[k]['avg_launch_ms'], r[k]['frac']) for k in ('sweep_A','sweep_C','sweep_E','advect_stage')})"
done
