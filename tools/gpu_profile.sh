#!/bin/bash
# tools/gpu_profile.sh <tag> -- run on the GPU box (through gpurun) from the repo root:
#   1. (optional) pytest -m gpu   2. bench.py   3. rocprofv3 kernel stats of the same bench command
#   4. separate PMC passes (FETCH_SIZE / WRITE_SIZE) -- no trace domains mixed in (gpurun rule)
# Everything lands under gpurun_out/; tools/prof_summary.py <tag> turns it into profiles/<tag>_*.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-r01}
STEPS=${STEPS:-2}
nproc > $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt
if [ "${RUN_TESTS:-0}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -14 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json
BENCH="python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stats -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
# the raw per-dispatch trace is large: keep only what the summary needs from it -- per kernel, the launches that did
# work (a BiCGSTAB sweep enqueued behind a finished solve returns at once; those are not part of the per-launch average)
python - "$OUT/prof_$TAG" <<'PY'
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
du -sh $OUT
