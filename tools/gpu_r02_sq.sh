#!/bin/bash
# SQ counters of the step's kernels in two separate PMC passes (no trace domains mixed in): where a wave's cycles go.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-r02}
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg --no-kernel-timers --no-verify"
rm -rf $OUT/sq1_$TAG $OUT/sq2_$TAG
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1_$TAG -o pmc -- $BENCH > $OUT/sq1_$TAG.log 2>&1; echo "sq1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/sq2_$TAG -o pmc -- $BENCH > $OUT/sq2_$TAG.log 2>&1; echo "sq2 rc=$?"
python - $TAG <<'PY'
import csv, glob, collections, sys, json
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("sq1", "sq2"):
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:34]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:9]:
    a = acc[k]; n = cnt[k]["SQ_WAVE_CYCLES"] or 1; wc = a.get("SQ_WAVE_CYCLES", 1) or 1
    row = {"launches": n, "wave_cycles_per_launch": wc / n, "wait_any": a["SQ_WAIT_ANY"] / wc, "wait_inst_any": a["SQ_WAIT_INST_ANY"] / wc,
           "active_inst_any": a["SQ_ACTIVE_INST_ANY"] / wc, "active_valu": a["SQ_ACTIVE_INST_VALU"] / wc, "active_lds": a["SQ_ACTIVE_INST_LDS"] / wc,
           "active_vmem": a["SQ_ACTIVE_INST_VMEM"] / wc, "busy_cycles_per_launch": a["SQ_BUSY_CYCLES"] / n}
    n2 = cnt[k]["SQ_INSTS_VALU"] or 1
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_MFMA"):
        row[c.lower() + "_per_launch"] = a[c] / n2
    out[k] = row
    print(k, json.dumps({x: (round(y, 3) if y < 100 else round(y)) for x, y in row.items()}))
json.dump(out, open("gpurun_out/%s_sq_counters.json" % tag, "w"), indent=1)
PY
