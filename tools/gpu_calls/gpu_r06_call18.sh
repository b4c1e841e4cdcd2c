#!/bin/bash
# round 6, call 18: the two-launch organisation on the hybrid operator: tile statistics, kernel durations of both forms
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CUP2D_HOST_TIMING=1 FORM=auto LFINE=9 NOTIMING=1 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "install_sell: .* tiles|AMR step" | head -3
for F in auto full; do
  rm -rf /tmp/prof_h
  FORM=$F LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o stats -- python3 tools/gpu_amr_bench.py > /tmp/prof_h.log 2>&1
  grep "AMR step" /tmp/prof_h.log
  f=$(find /tmp/prof_h -name "stats_kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n = r["Name"].replace("cup2d::", "").replace("void ", ""); n = n[:n.find("(")] if "(" in n else n
    print("  %-40s %6s calls  avg %8.1f us  min %8.1f  max %8.1f  %5.1f%%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
done
