#!/bin/bash
# fused-solver timing + FETCH/WRITE traffic of the solver kernels in one call
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SKIP_CHECK=1 VARIANTS=fused1 timeout 120 python tools/gpu_variants.py 2>&1 | grep -E "^TIME|rror"
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/q_$c
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/q_$c -o pmc -- $BENCH > $OUT/q_$c.log 2>&1
done
python - <<'PY'
import csv, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/q_%s/pmc_counter_collection.csv" % c)):
        acc[r["Kernel_Name"].split("(")[0].replace("void cup2d::", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        full = [x for x in v if x >= 0.05 * max(v)]
        res[k][c] = sum(full) / len(full) * 1024 * (2 if c == "FETCH_SIZE" else 1)
for k, d in res.items():
    if k.startswith("k_fused") or k.startswith("k_sweep"):
        print("%-22s read %.1f B/cell  write %.1f B/cell" % (k, d.get("FETCH_SIZE", 0) / 16777216, d.get("WRITE_SIZE", 0) / 16777216))
PY
