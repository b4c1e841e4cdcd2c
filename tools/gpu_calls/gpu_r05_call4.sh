#!/bin/bash
# round 5, call 4: the whole GPU suite on the pruned library (stored ring, ghost-edge form, previous-round exports, fifteen
# environment tunables gone), the per-iteration timeline of the doubly periodic patch (deferred N-rank organisation), the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/r05c4_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r05c4_pytest.log | tail -12
grep -n "first worker traceback" -A 25 $OUT/r05c4_pytest.log | cut -c1-300 | head -60
grep -n "Error\|assert " $OUT/r05c4_pytest.log | cut -c1-300 | head -30
cd /tmp
rm -rf /tmp/prof_xy
AXES=xy STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_xy -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r05c4_self_xy.log 2>&1
grep -v "^W2026\|^E2026\|simple_timer" $GRAFT_REPO_ROOT/$OUT/r05c4_self_xy.log | tail -8 | cut -c1-700
f=$(find /tmp/prof_xy -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r05_nrank_timeline.txt
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05c4_bench.json 2> $OUT/r05c4_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05c4_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "verified_summary") if k in d})
    print("roofline", d.get("roofline"))
    print("placement", {k: v for k, v in (d.get("placement") or {}).items() if k != "what"})
    print("tolerance", json.dumps(d.get("solve_to_tolerance"))[:1200])
    print("second_size", d.get("second_size"))
    n = d.get("nrank_path_on_one_gpu") or {}
    print("nrank", {k: n.get(k) for k in ("blocks", "ratio_to_plain", "fixed_us_per_iteration_over_plain", "ms_per_step", "solver_form")}, json.dumps(n.get("solve_to_tolerance"))[:600])
    for o in n.get("other_patches", []):
        print("   ", {k: o.get(k) for k in ("blocks", "ghost_sides", "ratio_to_plain", "fixed_us_per_iteration_over_plain", "ms_per_step", "plain_context_ms_per_step", "error")})
    a = d.get("amr_configs4") or {}
    print("amr", {k: a.get(k) for k in ("value", "ms_per_step", "error")}, json.dumps(a.get("solve_to_tolerance"))[:500])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -5 $OUT/r05c4_bench.err | cut -c1-300
