#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/r02_bench32.json 2> $OUT/r02_bench32.err; echo "bench rc=$?"; tail -3 $OUT/r02_bench32.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench32.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"]["ok"]); print(json.dumps(d["amr_configs4"])[:1200])
PY
