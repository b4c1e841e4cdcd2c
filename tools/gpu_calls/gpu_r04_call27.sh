#!/bin/bash
# round 4, call 27: the gate on the round's last code (GPU suite with the driver's command line, smoke(), the bench line) and the
# kernel statistics of the block-AMR leg (rocprofv3 --kernel-trace --stats of tools/gpu_amr_bench.py at LFINE=9)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_final_gate.sh 2>&1 | tail -14
grep -E "amr_big|cup2d_run_mpi -levelMax" $OUT/gate_pytest.log | head -6
rm -rf $OUT/prof_r04amr
LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04amr -o stats -- python3 tools/gpu_amr_bench.py > $OUT/prof_r04amr.log 2>&1
echo "rocprof amr rc=$?"; grep "AMR step" $OUT/prof_r04amr.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r04amr/**/stats_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/r04_amr_kernel_stats.txt", "w") as o:
        o.write("# LFINE=9 NOTIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/gpu_amr_bench.py   (MI355X, gfx950)\n")
        o.write("# 63 412 blocks (three levels, finest 4096^2-equivalent), 2 + 5 steps of 50 BiCGSTAB iterations on the hybrid operator; ns\n")
        o.write("%-60s %7s %14s %12s %10s %10s %7s\n" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct"))
        for r in rows:
            n = r["Name"].replace("cup2d::", "").replace("void ", "")
            n = n[:n.find("(")] if "(" in n else n
            o.write("%-60s %7s %14s %12.0f %10s %10s %6.2f%%\n" % (n[:60], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], float(r["Percentage"])))
    print(open("gpurun_out/r04_amr_kernel_stats.txt").read()[:2500])
PY
find $OUT/prof_r04amr -name "*kernel_trace.csv" -delete
