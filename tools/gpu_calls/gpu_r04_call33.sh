#!/bin/bash
# round 4, call 33: cup2d_step enqueues the projection behind the solve's last pass before it waits for the solve to end
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_comm.py tests/test_amr.py tests/test_distributed.py -x -q -m gpu -p no:cacheprovider -k "step or steps or communicator or driver or run_with" > $OUT/r04c33_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c33_pytest.log | tail -5
for n in 2048 4096; do
  timeout 300 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-nrank-proxy 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n', d['config'].get('workload','')[:12], 'value', d['value'], 'ms', d['ms_per_step'], 'no-timers ms', d.get('ms_per_step_no_kernel_timers'), 'gpu ms', round(d['gpu_ms_per_step']['solver_sweeps']+d['gpu_ms_per_step']['outside_the_sweeps'],3), 'ok', d['verified']['ok'])"
done
