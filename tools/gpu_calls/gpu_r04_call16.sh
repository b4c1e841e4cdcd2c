#!/bin/bash
# round 4, call 16: z edges of the side two consecutive rounds share taken from the previous round's exports (CUP2D_EDGE_PREV, bit
# mask by MODE: 8 = C+D', 4 = E+A+B): parity against the five sweeps, then the step time per setting (three processes each)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for pv in 12 8; do
  CUP2D_EDGE_PREV=$pv timeout 900 python3 -m pytest tests/test_solver_variants_gpu.py -q -m gpu -p no:cacheprovider -k "forms_of_the_fused and eab" 2>&1 | grep -E "passed|failed|FAILED|assert|Error" | tail -4
done
run() {
  for rep in 1 2 3; do
    env "$@" python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$*  %.3f ms/step  C+D %.1f us  E+A+B %.1f us  verified %s' % (d['ms_per_step'], 1e3*k['sweep_C']['ms_avg'], 1e3*k['sweep_EA']['ms_avg'], d['verified']['ok']))"
  done
}
run CUP2D_EDGE_PREV=0
run CUP2D_EDGE_PREV=8
run CUP2D_EDGE_PREV=4
run CUP2D_EDGE_PREV=12
run CUP2D_EDGE_PREV=8 CUP2D_EDGE_SHARE=13
run CUP2D_EDGE_PREV=0
