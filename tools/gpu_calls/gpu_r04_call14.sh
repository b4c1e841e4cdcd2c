#!/bin/bash
# round 4, call 14: what kind of box is this (clocks, power cap, partition modes) next to its step time -- the two timing modes are per box
set -u
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showmemorypartition --showcomputepartition --showtemp 2>&1 | grep -v "^$" | head -60
python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-verify --no-nrank-proxy 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('ms/step %.3f  C+D %.1f us  E+A+B %.1f us  advect %.1f us' % (d['ms_per_step'], 1e3*k['sweep_C']['ms_avg'], 1e3*k['sweep_EA']['ms_avg'], 1e3*k['advect_stage']['ms_avg']))"
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head
