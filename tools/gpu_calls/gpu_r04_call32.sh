#!/bin/bash
# round 4, call 32: x = x0 + P_inv y_opt launched with the choice of y_opt read on the device (one host wait per solve instead
# of two): the solver tests, and the step at 2048^2 / 4096^2
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_solver_variants_gpu.py tests/test_spmat_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/r04c32_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c32_pytest.log | tail -5
for n in 2048 4096; do
  timeout 300 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n', d['config'].get('workload','')[:20], 'value', d['value'], 'ms', d['ms_per_step'], 'no-timers ms', d.get('ms_per_step_no_kernel_timers'), 'final_x', d['gpu_ms_per_step']['families'].get('final_x'), 'ok', d['verified']['ok'])"
done
