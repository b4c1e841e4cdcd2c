#!/bin/bash
# round 6, call 9: the next step's max|u| reduced and copied behind the projection (no launch, no wait at the start of a step);
# the placement search that goes on while all sets look alike; step tests, bench line, the step as the GPU sees it
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes_gpu.py tests/test_solver_variants_gpu.py tests/test_spmat_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
CUP2D_HOST_TIMING=1 python3 bench.py --steps 20 --warmup 5 > $OUT/r06c9_bench.json 2> $OUT/r06c9_bench.err; echo "bench rc=$?"
grep "tune_placement: set" $OUT/r06c9_bench.err | head -30 | cut -c1-160
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06c9_bench.json")); S = d["summary"]
print(d["value"], d["ms_per_step"], d["verified_ok"], S["placement"]); print(S["gpu_ms_per_step"], S["second_size_2048"], S["solve_to_tolerance"]); print(json.dumps(S["north_star"])); print(json.dumps(S["nrank_path_on_one_gpu"])); print(S["amr_configs4"])
PY
cd /tmp; rm -rf /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_step_timeline.py $f 2>&1 | head -24 | tee $GRAFT_REPO_ROOT/$OUT/r06_4096_step_timeline.txt
