#!/bin/bash
# round 4, call 17: what the contiguous walk costs by itself (no exports taken)
set -u
export TMPDIR=/tmp
run() {
  for rep in 1 2; do
    env "$@" python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy --no-verify 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$*  %.3f ms/step  C+D %.1f us  E+A+B %.1f us' % (d['ms_per_step'], 1e3*k['sweep_C']['ms_avg'], 1e3*k['sweep_EA']['ms_avg']))"
  done
}
run CUP2D_EDGE_WALK=0
run CUP2D_EDGE_WALK=8
run CUP2D_EDGE_WALK=12
run CUP2D_EDGE_PREV=12
run CUP2D_EDGE_WALK=0
