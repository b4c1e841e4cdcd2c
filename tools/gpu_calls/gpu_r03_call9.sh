#!/bin/bash
# grouped host look on the N-rank finish paths (merge 2): ranks sharing the GPU (callbacks), one rank through RCCL; --force-dist timing
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_distributed.py tests/test_comm.py tests/test_solver_variants_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/r03_call9_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r03_call9_pytest.log
for g in 1 4; do
  CUP2D_SOLVE_GROUP=$g timeout 300 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist group $g', d['value'], d['ms_per_step'], d['config']['comm'].get('selftest'))"
done
timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-verify --no-kernel-timers 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'])"
