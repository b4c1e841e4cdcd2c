#!/bin/bash
# the decomposed code path on one rank (in-library communicator, MERGE 2) against the plain path, both in the default organisation
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_forcedist.jsonl
for extra in "" "--force-dist"; do
  timeout 600 python3 bench.py --gpus 1 $extra --steps 10 --warmup 3 --no-cpu-baseline --no-amr 2>/dev/null | tail -1 >> $OUT/r03_forcedist.jsonl
done
python3 - <<'PY'
import json
for line in open("gpurun_out/r03_forcedist.jsonl"):
    d = json.loads(line)
    ra = d["roofline_all"]
    print(d["config"]["parallelism"], d["value"], d["ms_per_step"], {k: (ra[k]["kernel"], ra[k]["avg_launch_ms"], ra[k]["frac"]) for k in ra if k.startswith("sweep")}, d["verified"]["ok"], d.get("comm", {}).get("selftest"))
PY
