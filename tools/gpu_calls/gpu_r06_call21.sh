#!/bin/bash
# round 6, call 21: k_hyb_rows with one record per list entry, requested in front of the status check
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_amr.py tests/test_spmat_gpu.py -x -q -m gpu -p no:cacheprovider -k "tile_fused_solver or installed_from_the_tables or spmat or hybrid or matrix" > $OUT/c21_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error|assert " $OUT/c21_pytest.log | cut -c1-250 | tail -8
for F in auto; do
  rm -rf /tmp/prof_h
  FORM=$F LFINE=9 NOTIMING=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o stats -- python3 tools/gpu_amr_bench.py > /tmp/prof_h.log 2>&1
  grep -E "AMR step" /tmp/prof_h.log | head -4
  f=$(find /tmp/prof_h -name "stats_kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    n = r["Name"].replace("cup2d::", "").replace("void ", ""); n = n[:n.find("(")] if "(" in n else n
    print("  %-40s %6s calls  avg %8.1f us  min %8.1f  max %8.1f  %5.1f%%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
done
FORM=auto LFINE=9 NOTIMING=1 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "AMR step"
