#!/bin/bash
# round 6, the gate on the round's code, as the driver runs it: the GPU suite (driver's command line + durations), smoke(), the
# bench line; then the profile set of the same bench command (rocprofv3 kernel stats, FETCH / WRITE passes: separate), the
# N-rank timelines (three patches), the adapted-grid kernel statistics, the step as the GPU sees it
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 > $OUT/gate_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|^[0-9.]+s call" $OUT/gate_pytest.log | tail -22
grep -n "first worker traceback" -A 25 $OUT/gate_pytest.log | cut -c1-300 | head -50
grep -n "Error\|assert " $OUT/gate_pytest.log | cut -c1-300 | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_r06.json 2> $OUT/bench_r06.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; wc -c $OUT/bench_r06.json; cat $OUT/bench_r06.json
cp $OUT/bench_detail.json $OUT/bench_r06_detail.json
tail -3 $OUT/bench_r06.err | cut -c1-300
# ---- profile set ----
TAG=r06; STEPS=3
BENCH="python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
CUP2D_BENCH_DETAIL=/tmp/d1.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stats -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
CUP2D_BENCH_DETAIL=/tmp/d2.json timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
CUP2D_BENCH_DETAIL=/tmp/d3.json timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
python3 - "$OUT/prof_$TAG" <<'PY'
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/stats_kernel_trace.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(d + "/stats_full_launches.csv", "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "FullCalls", "FullAvgNs", "FullMinNs", "FullMaxNs"])
    for k, v in acc.items():
        full = [x for x in v if x >= 0.05 * max(v)]
        w.writerow([k, len(v), len(full), sum(full) / len(full), min(full), max(full)])
PY
# the step as the GPU saw it (from the same trace), then drop the raw trace
f=$(find $OUT/prof_$TAG -name "stats_kernel_trace.csv" | head -1)
python3 tools/kernel_step_timeline.py $f "k_pressure_rhs" 3 2>&1 | head -24 | tee $OUT/r06_4096_step_timeline.txt
rm -f $f
STEPS=3 python3 tools/prof_summary.py $TAG 2>&1 | tail -4
# ---- N-rank timelines ----
cd /tmp
for P in "xy 512 512 r06_nrank_timeline" "xy 512 256 r06_nrank_timeline_configs3" "y 512 256 r06_nrank_timeline_configs3_long_sides" "x 512 512 r06_nrank_timeline_x"; do
  set -- $P
  rm -rf /tmp/prof_n
  AXES=$1 NBX=$2 NBY=$3 STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_n -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/$4.log 2>&1
  grep -E "ms/step|N-rank path" $GRAFT_REPO_ROOT/$OUT/$4.log | cut -c1-150
  f=$(find /tmp/prof_n -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/$4.txt | head -9
done
cd $GRAFT_REPO_ROOT
# ---- adapted grid ----
rm -rf $OUT/prof_r06amr
LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06amr -o stats -- python3 tools/gpu_amr_bench.py > $OUT/prof_r06amr.log 2>&1
echo "rocprof amr rc=$?"; grep "AMR step" $OUT/prof_r06amr.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r06amr/**/stats_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/r06_amr_kernel_stats.txt", "w") as o:
        o.write("# LFINE=9 NOTIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/gpu_amr_bench.py   (MI355X, gfx950)\n")
        o.write("# 63 412 blocks (three levels, finest 4096^2-equivalent), 2 + 5 steps of 50 BiCGSTAB iterations on the hybrid operator; ns\n")
        o.write("%-60s %7s %14s %12s %10s %10s %7s\n" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct"))
        for r in rows:
            n = r["Name"].replace("cup2d::", "").replace("void ", "")
            n = n[:n.find("(")] if "(" in n else n
            o.write("%-60s %7s %14s %12.0f %10s %10s %6.2f%%\n" % (n[:60], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], float(r["Percentage"])))
    print(open("gpurun_out/r06_amr_kernel_stats.txt").read()[:1500])
PY
find $OUT/prof_r06amr -name "*kernel_trace.csv" -delete
rocm-smi --showclocks 2>&1 | grep -E "fclk|mclk|sclk" | head -4
du -sh $OUT
