#!/bin/bash
# profile set of the round's final kernels (two-launch BiCGSTAB organisation) + size sweep + the decomposed path on one rank
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_profile_set.sh r03 2>&1 | tail -40
python3 tools/prof_summary.py r03 2>&1 | tail -5
rm -f $OUT/r03_size_sweep.jsonl
for n in 2048 4096 8192; do
  timeout 600 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr 2>/dev/null | tail -1 >> $OUT/r03_size_sweep.jsonl
done
timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-amr 2>/dev/null | tail -1 >> $OUT/r03_size_sweep.jsonl
python3 - <<'PY'
import json
for line in open("gpurun_out/r03_size_sweep.jsonl"):
    d = json.loads(line)
    ra = d["roofline_all"]
    print(d["config"]["workload"][:22], d["config"]["parallelism"], d["value"], d["ms_per_step"], {k: (ra[k]["avg_launch_ms"], ra[k]["frac"]) for k in ("sweep_C", "sweep_EA", "advect_stage") if k in ra}, d["verified"]["ok"])
PY
cp profiles/r03_*.json profiles/r03_*.txt $OUT/ 2>/dev/null
