#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_solver_variants_gpu.py tests/test_abi_and_host.py tests/test_amr.py tests/test_spmat_gpu.py -m gpu -q -x 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r02_bench40.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench40.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_no_kernel_timers"], d["verified"]["ok"], "amr", d["amr_configs4"]["value"], d["amr_configs4"]["ms_per_step"])
PY
