#!/bin/bash
# tools/gpu_profile.sh -- run on the GPU box (through gpurun) from the repo root:
#   1. pytest -m gpu   2. bench.py   3. rocprofv3 kernel stats of the same bench command
#   4. separate PMC passes (FETCH_SIZE / WRITE_SIZE) -- no trace domains mixed in (gpurun rule)
# Everything lands under gpurun_out/; summaries are copied to profiles/ by tools/prof_summary.py.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-r01}
STEPS=${STEPS:-2}
nproc > $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> $OUT/host.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json
BENCH="python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stats -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
find $OUT -name '*.csv' | head -20
# keep only what is needed (64 MiB merge limit): drop the raw kernel trace if huge
du -sh $OUT
