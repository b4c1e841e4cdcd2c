#!/bin/bash
# round 4, call 19: the placement search of the solver's vectors: five fresh bench processes with it, three without; solver tests
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() {
  for rep in $(seq 1 $1); do
    shift_args="${@:2}"
    env $shift_args python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>$OUT/r04c19.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; p=d.get('placement') or {}
print('$shift_args  %.1f Mcell-updates/s  %.3f ms/step  C+D %.1f us  E+A+B %.1f us  verified %s  placement: %s sets, kept %.1f / slowest %.1f / first %.1f us' % (d['value'], d['ms_per_step'], 1e3*k['sweep_C']['ms_avg'], 1e3*k['sweep_EA']['ms_avg'], d['verified']['ok'], p.get('candidates'), p.get('kept_us',0), p.get('slowest_us',0), p.get('first_us',0)))"
  done
}
run 6 CUP2D_PLACEMENT_TRIES=16
run 2 CUP2D_PLACEMENT_TRIES=0
run 4 CUP2D_PLACEMENT_TRIES=32
CUP2D_HOST_TIMING=1 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-verify 2>&1 >/dev/null | grep "tune_placement" | tail -12
echo
