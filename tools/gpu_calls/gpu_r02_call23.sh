#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
LFINE=9 timeout 600 python tools/gpu_amr_bench.py > $OUT/r02_amr25.log 2>&1; tail -30 $OUT/r02_amr23.log
cd /tmp && LFINE=9 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_amr25 -o amr -- python $GRAFT_REPO_ROOT/tools/gpu_amr_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $OUT/prof_amr25 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-90s calls %6s avg %9.1f us tot %6.2f %%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
