#!/bin/bash
# smoother kernels: kernel-trace durations + FETCH/WRITE traffic (+ optional extra counters: CTRS="A B")
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/gpu_smoother.py"
rm -rf $OUT/sm_stats; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sm_stats -o s -- $CMD > $OUT/sm_stats.log 2>&1
grep -h "k_smoother\|k_laplacian" $OUT/sm_stats/s_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE ${CTRS:-}; do
  rm -rf $OUT/sm_$c
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/sm_$c -o pmc -- $CMD > $OUT/sm_$c.log 2>&1
done
rm -f $OUT/sm_stats/s_kernel_trace.csv
python - <<'PY'
import csv, collections, os
n = 4096 * 4096
for c in ["FETCH_SIZE", "WRITE_SIZE"] + os.environ.get("CTRS", "").split():
    f = "gpurun_out/sm_%s/pmc_counter_collection.csv" % c
    if not os.path.exists(f):
        print(c, "missing"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void cup2d::", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k.startswith("k_smoother") or k.startswith("k_laplacian"):
            m = sum(v) / len(v)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                print("%-26s %-12s %.2f B/cell" % (k, c, m * 1024 * (2 if c == "FETCH_SIZE" else 1) / n))
            else:
                print("%-26s %-12s %.4g" % (k, c, m))
PY
