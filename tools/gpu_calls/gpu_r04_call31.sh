#!/bin/bash
# round 4, call 31: the bench line after the AMR leg learned to report the 8-rank plan (default flags, as the driver runs it)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
python3 bench.py > $OUT/r04c31_bench.json 2> $OUT/r04c31_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04c31_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"]["ok"], d["cpu_baseline"]["value"], d["amr_configs4"]["value"], d["amr_configs4"]["plan_on_8_ranks"])
print({k: d["nrank_path_on_one_gpu"][k] for k in list(d["nrank_path_on_one_gpu"])[:6]})
PY
tail -3 $OUT/r04c31_bench.err
