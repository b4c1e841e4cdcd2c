#!/bin/bash
# round 5, the gate on the round's code, as the driver runs it: the GPU suite (driver's command line + durations), smoke(), the
# bench line; then the profile set of the same bench command (rocprofv3 kernel stats, FETCH / WRITE passes: separate)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 > $OUT/gate_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s call" $OUT/gate_pytest.log | tail -32
grep -n "first worker traceback" -A 25 $OUT/gate_pytest.log | cut -c1-300 | head -50
grep -n "Error\|assert " $OUT/gate_pytest.log | cut -c1-300 | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r05.json 2> $OUT/bench_r05.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_r05.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "verified_summary") if k in d})
    print("roofline", d.get("roofline"))
    print("placement", {k: v for k, v in (d.get("placement") or {}).items() if k != "what"})
    n = d.get("roofline_north_star") or {}
    print("north", {k: n.get(k) for k in ("avg_launch_ms", "frac")}, json.dumps(n.get("floors_on_this_box_us"))[:700])
    t = d.get("solve_to_tolerance") or {}
    print("tolerance", {k: t.get(k) for k in ("iters_median", "ms_per_step_median", "ms_not_in_iterations_or_fringe", "error")})
    print("second_size", d.get("second_size"))
    nr = d.get("nrank_path_on_one_gpu") or {}
    print("nrank", {k: nr.get(k) for k in ("blocks", "ratio_to_plain", "fixed_us_per_iteration_over_plain", "solver_form", "error")})
    for o in nr.get("other_patches", []):
        print("   ", {k: o.get(k) for k in ("blocks", "ghost_sides", "ratio_to_plain", "fixed_us_per_iteration_over_plain", "error")})
    a = d.get("amr_configs4") or {}
    print("amr", {k: a.get(k) for k in ("value", "ms_per_step", "error")}, (a.get("regrid") or {}).get("ms"))
    print("gpu_ms", {k: v for k, v in (d.get("gpu_ms_per_step") or {}).items() if k != "families" and k != "note"})
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 $OUT/bench_r05.err | cut -c1-300
# ---- profile set (tools/gpu_profile.sh without its own full bench run) ----
TAG=r05; STEPS=3
BENCH="python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stats -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
python3 - "$OUT/prof_$TAG" <<'PY'
import collections, csv, sys
d = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(d + "/stats_kernel_trace.csv")):
    acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(d + "/stats_full_launches.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "FullCalls", "FullAvgNs", "FullMinNs", "FullMaxNs"])
    for k, v in acc.items():
        full = [x for x in v if x >= 0.05 * max(v)]
        w.writerow([k, len(v), len(full), sum(full) / len(full), min(full), max(full)])
PY
rm -f $OUT/prof_$TAG/stats_kernel_trace.csv
cp $OUT/bench_r05.json $OUT/bench_$TAG.json
STEPS=3 python3 tools/prof_summary.py $TAG 2>&1 | tail -12
rocm-smi --showclocks 2>&1 | grep -E "fclk|mclk|sclk" | head -4
du -sh $OUT
