#!/bin/bash
# round 6, the last commit as the driver runs it: the GPU suite, smoke, the bench line (run again on the second half of the round: placement repair, hand-over in C+D', k_edge HYB on request)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/final_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/final_pytest.log | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
CUP2D_HOST_TIMING=1 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; wc -c $OUT/final_bench.json; cat $OUT/final_bench.json
cp $OUT/bench_detail.json $OUT/final_bench_detail.json
# the step as the GPU sees it
cd /tmp; rm -rf /tmp/prof_s
CUP2D_BENCH_DETAIL=/tmp/d.json timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify > /dev/null 2>&1
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_step_timeline.py $f "k_pressure_rhs" 3 2>&1 | head -24 | tee $GRAFT_REPO_ROOT/$OUT/r06_4096_step_timeline.txt
cd $GRAFT_REPO_ROOT
# ---- adapted grid: kernel statistics on the round's last code (the quads of an adapted grid take the quad WENO5 kernel) ----
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
    print(open("gpurun_out/r06_amr_kernel_stats.txt").read()[:1200])
PY
find $OUT/prof_r06amr -name "*kernel_trace.csv" -delete
