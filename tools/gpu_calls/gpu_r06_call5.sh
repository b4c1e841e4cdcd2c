#!/bin/bash
# round 6, call 5: projection writes the velocity non-temporally (RK stage 1 of the next step no longer pays its write-back); the
# whole GPU suite; the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python3 tools/gpu_stage1_conditions.py 2>&1 | tail -6
python3 bench.py --steps 20 --warmup 5 > $OUT/r06c5_bench.json 2> $OUT/r06c5_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06c5_bench.json")); S = d["summary"]
print(d["value"], d["ms_per_step"], d["verified_ok"]); print(json.dumps(S["north_star"])); print(json.dumps(S["kernels"])); print(S["gpu_ms_per_step"], S["second_size_2048"], S["amr_configs4"]["value"])
PY
cp gpurun_out/bench_detail.json $OUT/r06c5_bench_detail.json
timeout 1500 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/r06c5_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $OUT/r06c5_gpu_tests.log | cut -c1-300
