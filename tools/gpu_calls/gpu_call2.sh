#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/gpu_variants.py > $OUT/variants2.log 2>&1; echo "variants rc=$?"; grep -E "CHECK|TIME|VARIANTS|rror" $OUT/variants2.log | tail -12
CUP2D_PRECOND=mfma VARIANTS=sweeps0 timeout 120 python tools/gpu_variants.py > $OUT/variants_mfma.log 2>&1; echo "MFMA-precond sweeps: $(grep -E "^CHECK|^TIME" $OUT/variants_mfma.log)"
for d in 1 2 4 6 7; do
  CUP2D_FUSED_DBG=$d SKIP_CHECK=1 VARIANTS=fused0 timeout 120 python tools/gpu_variants.py > $OUT/variants_dbg$d.log 2>&1
  echo "DBG=$d: $(grep -E '^TIME' $OUT/variants_dbg$d.log)"
done
timeout 200 python -m pytest tests/test_solver_variants_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_variants.log 2>&1; echo "pytest variants rc=$?"; tail -4 $OUT/pytest_variants.log
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --solver fused --finish launch"
rm -rf $OUT/pmc_fetch_fused $OUT/pmc_write_fused
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_fused -o pmc -- $BENCH > $OUT/pmc_fetch_fused.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_fused -o pmc -- $BENCH > $OUT/pmc_write_fused.log 2>&1; echo "pmc write rc=$?"
python - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write"):
    fs = glob.glob("gpurun_out/pmc_%s_fused/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        print(tag, "%-60s launches=%d avg_KiB=%.0f" % (k, n, v / n))
PY
du -sh $OUT
