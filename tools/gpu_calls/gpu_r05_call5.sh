#!/bin/bash
# round 5, call 5: the placement experiment (arenas carved at a stride of 2^27 + pad against separate allocations, three fresh
# processes), the in-place test after its trim, the bench line with the north-star floors measured on the box
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TRIALS=3 timeout 400 python3 tools/gpu_placement_arena.py 4096 2>&1 | tee $OUT/r05_placement_arena.txt | cut -c1-200
t0=$(date +%s)
timeout 600 python3 -m pytest tests/test_comm.py -q -m gpu -p no:cacheprovider -k "received_in_place" > $OUT/r05c5_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r05c5_pytest.log | tail -5
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05c5_bench.json 2> $OUT/r05c5_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05c5_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step") if k in d}, (d.get("verified_summary") or {}).get("ok"))
    print("placement", {k: v for k, v in (d.get("placement") or {}).items() if k != "what"})
    n = d.get("roofline_north_star") or {}
    print("north", {k: n.get(k) for k in ("avg_launch_ms", "frac", "stage1", "stage2")})
    print("floors", json.dumps(n.get("floors_on_this_box_us"))[:900])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 $OUT/r05c5_bench.err | cut -c1-300
