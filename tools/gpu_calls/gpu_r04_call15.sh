#!/bin/bash
# round 4, call 15: exactly max_iter iterations enqueued (no launches behind the cap): solver tests, 2048^2 / 4096^2, N-rank path
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_comm.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for n in 2048 4096; do
python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('n=%d  %.1f Mcell-updates/s  ms/step %.3f  C+D %.1f us x%d  E+A+B %.1f us  verified %s' % ($n, d['value'], d['ms_per_step'], 1e3*k['sweep_C']['ms_avg'], k['sweep_C']['launches'], 1e3*k['sweep_EA']['ms_avg'], d['verified']['ok']))"
done
python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "plain|self-periodic \(|N-rank" | tail -4
NBX=512 NBY=256 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "plain|self-periodic \(|N-rank" | tail -4
LFINE=9 NOTIMING=1 python3 tools/gpu_amr_bench.py 2>&1 | grep "AMR step"
rocm-smi --showclocks 2>&1 | grep -E "fclk|mclk" | head -3
