#!/bin/bash
# one gpurun call: variants A/B first (most informative), then the new solver tests, then the rest of -m gpu,
# then a kernel-stats profile of the fused configuration.  Everything is bounded by its own timeout.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt
timeout 300 python tools/gpu_variants.py > $OUT/variants.log 2>&1; echo "variants rc=$?"; grep -E "CHECK|TIME|VARIANTS|Error|error" $OUT/variants.log | tail -20
timeout 400 python -m pytest tests/test_solver_variants_gpu.py -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest_variants.log 2>&1; echo "pytest variants rc=$?"; tail -25 $OUT/pytest_variants.log
timeout 700 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider --deselect tests/test_solver_variants_gpu.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rest rc=$?"; tail -25 $OUT/pytest_gpu.log
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --solver fused --finish kernel"
rm -rf $OUT/prof_fused
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fused -o stats -- $BENCH > $OUT/prof_fused.log 2>&1; echo "rocprof stats rc=$?"
rm -f $OUT/prof_fused/*kernel_trace.csv $OUT/prof_fused/*/*kernel_trace.csv
find $OUT/prof_fused -name "*kernel_stats.csv" | head -1 | xargs -r head -12
du -sh $OUT
