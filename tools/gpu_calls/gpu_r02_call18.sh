#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_solver_variants_gpu.py tests/test_gpu_parity.py tests/test_distributed.py tests/test_comm.py tests/test_baseline_sizes_gpu.py -m gpu -q -k "not functor" > $OUT/r02_pytest18.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r02_pytest18.log
VARIANTS=fused1 SKIP_CHECK=1 python tools/gpu_variants.py 2>&1 | grep TIME
python bench.py --no-cpu-baseline --steps 10 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step_no_kernel_timers'], d['verified']['ok'])"
python bench.py --gpus 2; echo "bench --gpus 2 on a 1-GPU box: rc=$?"
