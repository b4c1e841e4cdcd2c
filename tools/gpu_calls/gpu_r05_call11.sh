#!/bin/bash
# round 5, call 11: sparser event sampling in the timed region (every 16th iteration, outer launches every 4th step): the line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for args in "--steps 20 --warmup 5" ""; do
  t0=$(date +%s)
  timeout 900 python3 bench.py $args --no-cpu-baseline --no-amr --no-nrank-proxy --no-north-star-floors > $OUT/r05c11_bench.json 2> $OUT/r05c11_bench.err
  echo "bench [$args] rc=$? ($(( $(date +%s) - t0 )) s)"
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c11_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_no_kernel_timers", "steps", "roofline_extra_sampled_steps_outside_timed_region")}, d["verified_summary"]["ok"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_ms", "launches")})
print({k: (v["launches"], v["ms_avg"]) for k, v in d["kernels"].items() if v["launches"]})
print("placement", {k: round(v, 1) for k, v in d["placement"].items() if k != "what"})
PY
done
timeout 600 python3 -m pytest tests/test_bench_world8.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
