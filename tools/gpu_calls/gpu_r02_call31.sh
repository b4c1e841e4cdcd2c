#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r02_bench31.json 2> $OUT/r02_bench31.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench31.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"]["ok"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"])
for k in d.get("kernels",{}): pass
print({k:(v.get("avg_launch_ms"),v.get("frac")) for k,v in d.get("rooflines",{}).items()} if "rooflines" in d else list(d.keys()))
PY
timeout 900 python -m pytest tests/test_amr.py -m gpu -q -k "cpp_host" 2>&1 | tail -2
