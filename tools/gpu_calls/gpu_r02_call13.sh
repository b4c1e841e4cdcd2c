#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "TCC_[A-Z0-9_]*(WRITE|WRREQ|READ|RDREQ|REQ)[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' | head -c 3000; echo
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-verify"
for set in "TCC_READ_sum TCC_WRITE_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" ; do
rm -rf $OUT/tccx; timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/tccx -o pmc -- $BENCH > $OUT/tccx.log 2>&1; echo "rc=$? ($set)"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/tccx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "")[:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in ("k_sweepE_y<1>", "k_fused<0, 1>", "k_fused<1, 1>", "k_zero", "k_reduce_partial<0>"):
    if k in acc: print(k, {c: round(v / cnt[k][c] / 1e6, 3) for c, v in acc[k].items()}, "M per launch")
PY
done
