#!/bin/bash
# round 5, the last call: the driver's sequence on the round's last code -- the GPU suite (driver's command line), smoke(), the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $OUT/final_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s call" $OUT/final_pytest.log | tail -16
grep -n "first worker traceback" -A 25 $OUT/final_pytest.log | cut -c1-300 | head -50
grep -n "Error\|assert " $OUT/final_pytest.log | cut -c1-300 | head -12
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python3 bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench (driver's defaults) rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup", "verified_summary") if k in d})
    print("roofline", {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "frac", "avg_launch_ms", "traffic_source")})
    print("placement", {k: v for k, v in (d.get("placement") or {}).items() if k != "what"})
    print("cpu", {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "unit", "cores", "kind")})
    print(len(json.dumps(d)), "bytes; first keys:", list(d)[:9])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 $OUT/final_bench.err | cut -c1-300
