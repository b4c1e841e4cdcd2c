#!/bin/bash
# round 4, call 12: the bench line with the N-rank-path leg
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
python3 bench.py > $OUT/r04c12_bench.json 2> $OUT/r04c12_bench.err; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04c12_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["traffic_source"], d["roofline"]["frac"])
print(json.dumps(d["nrank_path_on_one_gpu"]))
print(d["amr_configs4"]["value"], d["amr_configs4"]["regrid"]["ms"], d["amr_configs4"]["regrid"]["first_ms"])
PY
tail -3 $OUT/r04c12_bench.err
