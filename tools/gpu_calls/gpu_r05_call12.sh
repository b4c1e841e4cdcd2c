#!/bin/bash
# round 5, call 12 (records): the rest of the profile set on the round's library -- SQ counters (two passes), L2 request counters
# (two passes), and the kernel statistics of the block-AMR leg (the AMR block operators take a block list since this round)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_r02_sq.sh r05 2>&1 | tail -12
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify"
TAG=r05
rm -rf $OUT/tcc_$TAG $OUT/tcc2_$TAG
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$TAG -o pmc -- $BENCH > $OUT/tcc_$TAG.log 2>&1; echo "tcc rc=$?"
timeout 300 rocprofv3 --pmc TCC_READ_sum TCC_WRITE_sum --output-format csv -d $OUT/tcc2_$TAG -o pmc -- $BENCH > $OUT/tcc2_$TAG.log 2>&1; echo "tcc2 rc=$?"
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("tcc", "tcc2"):
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(acc, key=lambda k: -acc[k]["TCC_REQ_sum"])[:12]:
    a = acc[k]
    row = {c.lower().replace("_sum", "") + "_per_launch": a[c] / max(1, cnt[k][c]) for c in a}
    row["hit_rate"] = a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
    out[k] = row
    print(k, {x: (round(y, 3) if y < 10 else round(y)) for x, y in row.items()})
json.dump(out, open("gpurun_out/%s_l2_requests.json" % tag, "w"), indent=1)
PY
rm -rf $OUT/prof_r05amr
LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r05amr -o stats -- python3 tools/gpu_amr_bench.py > $OUT/prof_r05amr.log 2>&1
echo "rocprof amr rc=$?"; grep "AMR step" $OUT/prof_r05amr.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r05amr/**/stats_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/r05_amr_kernel_stats.txt", "w") as o:
        o.write("# LFINE=9 NOTIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/gpu_amr_bench.py   (MI355X, gfx950)\n")
        o.write("# 63 412 blocks (three levels, finest 4096^2-equivalent), 2 + 5 steps of 50 BiCGSTAB iterations on the hybrid operator; ns\n")
        o.write("%-60s %7s %14s %12s %10s %10s %7s\n" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct"))
        for r in rows:
            n = r["Name"].replace("cup2d::", "").replace("void ", "")
            n = n[:n.find("(")] if "(" in n else n
            o.write("%-60s %7s %14s %12.0f %10s %10s %6.2f%%\n" % (n[:60], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], float(r["Percentage"])))
    print(open("gpurun_out/r05_amr_kernel_stats.txt").read()[:2200])
PY
find $OUT/prof_r05amr -name "*kernel_trace.csv" -delete
ls $OUT/*sq*r05* $OUT/r05_* 2>/dev/null | head
