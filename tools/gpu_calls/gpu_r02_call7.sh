#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_comm.py tests/test_distributed.py tests/test_solver_variants_gpu.py -m gpu -q --durations=4 > $OUT/r02_pytest7.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r02_pytest7.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 10 > $OUT/r02_bench7.json 2> $OUT/r02_bench7.err; echo "bench rc=$?"
timeout 600 python bench.py --force-dist --no-cpu-baseline --steps 10 > $OUT/r02_bench7_fd.json 2> $OUT/r02_bench7_fd.err; echo "bench force-dist rc=$?"
python - <<'PY'
import json
a = json.load(open("gpurun_out/r02_bench7.json")); b = json.load(open("gpurun_out/r02_bench7_fd.json"))
print("single %.1f (%.3f ms)  force-dist %.1f (%.3f ms)  ratio %.4f" % (a["value"], a["ms_per_step_no_kernel_timers"], b["value"], b["ms_per_step_no_kernel_timers"], b["ms_per_step_no_kernel_timers"] / a["ms_per_step_no_kernel_timers"]), b["config"]["comm"], a["verified"]["ok"], b["verified"]["ok"])
PY
done
