#!/bin/bash
# round 6, call 16: adapted grids, FAST arithmetic: the same-level quads through the quad WENO5 kernel (level by level), the rest
# through the per-block kernel: AMR tests, kernel statistics of the step, the bench's figure
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_amr.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for W in 1 0; do
  rm -rf $OUT/prof_q
  CUP2D_ADVECT_WALK=$W LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q -o stats -- python3 tools/gpu_amr_bench.py > $OUT/prof_q.log 2>&1
  echo "== CUP2D_ADVECT_WALK=$W"; grep "AMR step" $OUT/prof_q.log
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_q/**/stats_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r["Name"].replace("cup2d::", "").replace("void ", ""); n = n[:n.find("(")] if "(" in n else n
    if "advect" in n or "fillcases2" in n or "axpy" in n: print("   %-44s calls %5s  avg %8.1f us" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $OUT/prof_q -name "*kernel_trace.csv" -delete
done
