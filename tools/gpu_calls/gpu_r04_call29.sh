#!/bin/bash
# round 4, call 29 (run on several boxes): the headline line of a fresh process with the placement search, and what the box is
set -u
export TMPDIR=/tmp
for k in 1 2; do
  timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); ra=d['roofline_all']
print('value', d['value'], 'ms', d['ms_per_step'], 'C+D', round(1e3*ra['sweep_C']['avg_launch_ms'],1), 'E+A+B', round(1e3*ra['sweep_EA']['avg_launch_ms'],1), 'advect', round(1e3*ra['advect_stage']['avg_launch_ms'],1), 'placement', d['placement'], 'ok', d['verified']['ok'])"
done
rocm-smi --showserial --showuniqueid 2>&1 | grep -E "Serial|Unique" | head -2
