#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amr.py -m gpu -q -x -s -k "tile_fused" > $OUT/r02_pytest26a.log 2>&1; echo "fused hybrid rc=$?"; grep -E "blocks:|passed|failed|Error|assert" $OUT/r02_pytest26a.log | head -20
LFINE=9 timeout 600 python tools/gpu_amr_bench.py > $OUT/r02_amr26.log 2>&1; head -12 $OUT/r02_amr26.log
cd /tmp && LFINE=9 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_amr26 -o amr -- python $GRAFT_REPO_ROOT/tools/gpu_amr_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $OUT/prof_amr26 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]:
    print("%-90s calls %6s avg %9.1f us tot %6.2f %%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
