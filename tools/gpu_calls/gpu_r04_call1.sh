#!/bin/bash
# round 4, call 1: the new tests at configs[3]'s per-rank size (two ranks sharing the GPU), the 8192^2 solver test, a baseline bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_distributed.py tests/test_baseline_sizes_gpu.py -q -m gpu -p no:cacheprovider -s \
  -k "configs3 or 8192_two" > $OUT/r04c1_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "gpu_big|cup2d_run_mpi -n|8192\^2, 8|passed|failed|FAILED|Error|assert" $OUT/r04c1_pytest.log | tail -30
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04c1_bench.json 2> $OUT/r04c1_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04c1_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step") if k in d}, d.get("roofline"), d.get("verified"))
print(d.get("gpu_ms_per_step"))
PY
