#!/bin/bash
# round 6, second half: the profile set of the bench command on the round's last code (rocprofv3 kernel stats; FETCH / WRITE passes: separate),
# the N-rank timelines, the adapted-grid kernel statistics (both organisations)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
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
f=$(find $OUT/prof_$TAG -name "stats_kernel_trace.csv" | head -1)
rm -f $f
STEPS=3 python3 tools/prof_summary.py $TAG 2>&1 | tail -4
cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc_traffic.json $OUT/ 2>/dev/null
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
du -sh $OUT
