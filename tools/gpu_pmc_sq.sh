#!/bin/bash
# SQ stall breakdown of the solver kernels (separate PMC pass, no trace domains)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --solver ${SOLVER:-fused} --finish kernel"
rm -rf $OUT/pmc_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
tail -3 $OUT/pmc_sq.log
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:7]:
    a = acc[k]; wc = a.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-28s n=%d wave_cyc=%.3g wait_any=%.2f wait_inst=%.2f active=%.2f wait_lds=%.3f | lds_insts=%.3g bank_conflict_cyc=%.3g busy=%.3g"
          % (k, cnt[k]["SQ_WAVE_CYCLES"], wc, a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc,
             a["SQ_WAIT_INST_LDS"] / wc, a["SQ_INSTS_LDS"], a["SQ_LDS_BANK_CONFLICT"], a["SQ_BUSY_CYCLES"]))
PY
