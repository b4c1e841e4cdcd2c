#!/bin/bash
# round 6, call 49: is the 0.15 ms between the timed region (sampled events) and the same steps repeated without events the events or the order?
set -u
export TMPDIR=/tmp
F="--no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
for i in 1 2; do
  CUP2D_BENCH_DETAIL=/tmp/a.json timeout 300 python3 bench.py --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('events   value', d['value'], 'ms', d['ms_per_step'], 'repeat without', d['summary'].get('ms_per_step_no_kernel_timers'))"
  CUP2D_BENCH_DETAIL=/tmp/b.json timeout 300 python3 bench.py --steps 20 --warmup 5 $F --no-kernel-timers 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('noevents value', d['value'], 'ms', d['ms_per_step'], 'repeat without', d['summary'].get('ms_per_step_no_kernel_timers'))"
  CUP2D_BENCH_DETAIL=/tmp/c.json timeout 300 python3 bench.py --steps 20 --warmup 25 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('events w25 value', d['value'], 'ms', d['ms_per_step'], 'repeat without', d['summary'].get('ms_per_step_no_kernel_timers'))"
done
