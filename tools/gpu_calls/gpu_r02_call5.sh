#!/bin/bash
set -u
export TMPDIR=/tmp
bash tools/gpu_profile.sh r02a
bash tools/gpu_r02_sq.sh r02a
OUT=gpurun_out
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-verify"
rm -rf $OUT/tcc_r02a
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_r02a -o pmc -- $BENCH > $OUT/tcc_r02a.log 2>&1; echo "tcc rc=$?"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/tcc_r02a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "")[:34]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": cnt[k] += 1
for k in sorted(acc, key=lambda k: -acc[k]["TCC_REQ_sum"])[:8]:
    a = acc[k]; n = cnt[k] or 1
    print("%-36s launches %d  L2 req/launch %.3g  hit %.3g miss %.3g  hit rate %.3f" % (k, n, a["TCC_REQ_sum"] / n, a["TCC_HIT_sum"] / n, a["TCC_MISS_sum"] / n, a["TCC_HIT_sum"] / max(1, a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
PY
