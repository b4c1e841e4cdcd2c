#!/bin/bash
# one host look per group of BiCGSTAB iterations (CUP2D_SOLVE_GROUP): solver tests, body tests, A/B of the step time
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_spmat_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/r03_call7_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r03_call7_pytest.log
for g in 1 4 8; do
  for n in 4096 2048; do
    CUP2D_SOLVE_GROUP=$g timeout 300 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $g n $n', d['value'], d['ms_per_step'])"
  done
done
CUP2D_SOLVE_GROUP=1 timeout 300 python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amr group 1', d['amr_configs4'].get('value'), d['amr_configs4'].get('ms_per_step'))"
CUP2D_SOLVE_GROUP=4 timeout 300 python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amr group 4', d['amr_configs4'].get('value'), d['amr_configs4'].get('ms_per_step'))"
