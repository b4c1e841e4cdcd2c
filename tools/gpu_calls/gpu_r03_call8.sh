#!/bin/bash
# kernel-level view of the AMR step (63 k-block three-level grid): rocprofv3 kernel stats
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
LFINE=9 NOTIMING=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_amr_r03 -o stats -- python3 $GRAFT_REPO_ROOT/tools/gpu_amr_bench.py > $GRAFT_REPO_ROOT/$OUT/prof_amr_r03.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
grep -E "AMR step|grid:|operator" $OUT/prof_amr_r03.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_amr_r03/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n = r["Name"].replace("void cup2d::", "").replace("cup2d::", "")[:70]
    print("%-72s %6s calls  avg %8.1f us  %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
