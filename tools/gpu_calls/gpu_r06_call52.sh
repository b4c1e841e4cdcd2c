#!/bin/bash
# round 6, call 52 (records): the size sweep on one box with the round's last code
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SIDE="--no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
rm -f $OUT/r06_size_sweep.jsonl
for n in 2048 4096 8192; do
  CUP2D_BENCH_DETAIL=/tmp/d_$n.json timeout 600 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 $SIDE 2>/dev/null | tail -1 >> $OUT/r06_size_sweep.jsonl
done
CUP2D_BENCH_DETAIL=/tmp/d_fd.json timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 $SIDE 2>/dev/null | tail -1 >> $OUT/r06_size_sweep.jsonl
python3 - <<'PY'
import json
out = []
for line in open("gpurun_out/r06_size_sweep.jsonl"):
    d = json.loads(line); S = d["summary"]; ks = S["kernels"]
    row = {"workload": d["config"]["workload"][:24], "parallelism": d["config"]["parallelism"], "value": d["value"], "ms_per_step": d["ms_per_step"],
           "kernels_us_frac": {k: (ks[k]["us"], ks[k]["frac"]) for k in ("sweep_C", "sweep_EA", "advect_stage") if k in ks},
           "verified_ok": d["verified_ok"], "placement": S.get("placement")}
    out.append(row)
    print(row)
json.dump(out, open("gpurun_out/r06_size_sweep.json", "w"), indent=1)
PY
