#!/bin/bash
# profile set of a round (tools/gpu_profile_set.sh rNN): bench + rocprofv3 kernel stats + FETCH/WRITE passes (tools/gpu_profile.sh), SQ counters (two passes),
# L2 request counters -- every PMC pass separate, no trace domains mixed in.  tools/prof_summary.py r02 and this script's
# last step write the tracked summaries under profiles/.
set -u
export TMPDIR=/tmp
TAG=${1:-r03}
STEPS=3 bash tools/gpu_profile.sh $TAG
bash tools/gpu_r02_sq.sh $TAG
OUT=gpurun_out
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify"
rm -rf $OUT/tcc_$TAG $OUT/tcc2_$TAG
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$TAG -o pmc -- $BENCH > $OUT/tcc_$TAG.log 2>&1; echo "tcc rc=$?"
timeout 300 rocprofv3 --pmc TCC_READ_sum TCC_WRITE_sum --output-format csv -d $OUT/tcc2_$TAG -o pmc -- $BENCH > $OUT/tcc2_$TAG.log 2>&1; echo "tcc2 rc=$?"
python - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("tcc", "tcc2"):
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(acc, key=lambda k: -acc[k]["TCC_REQ_sum"])[:12]:
    a = acc[k]
    row = {c.lower().replace("_sum", "") + "_per_launch": a[c] / max(1, cnt[k][c]) for c in a}
    row["hit_rate"] = a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
    out[k] = row
    print(k, {x: (round(y, 3) if y < 10 else round(y)) for x, y in row.items()})
json.dump(out, open("gpurun_out/%s_l2_requests.json" % tag, "w"), indent=1)
PY
