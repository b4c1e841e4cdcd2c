#!/bin/bash
# round 4, call 9 (mid-round gate): the GPU suite with the driver's command line, smoke(), the bench line, then the profile
# set r04 (kernel stats, FETCH / WRITE traffic, SQ and L2 counters), the size sweep and the N-rank path on one rank
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_final_gate.sh 2>&1 | tail -12
bash tools/gpu_profile_set.sh r04 2>&1 | tail -30
python3 tools/prof_summary.py r04 2>&1 | tail -5
rm -f $OUT/r04_size_sweep.jsonl
for n in 2048 4096 8192; do
  timeout 600 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 --no-cpu-baseline --no-amr 2>/dev/null | tail -1 >> $OUT/r04_size_sweep.jsonl
done
timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-amr 2>/dev/null | tail -1 >> $OUT/r04_size_sweep.jsonl
python3 - <<'PY'
import json
for line in open("gpurun_out/r04_size_sweep.jsonl"):
    d = json.loads(line)
    ra = d["roofline_all"]
    print(d["config"]["workload"][:22], d["config"]["parallelism"], d["value"], d["ms_per_step"], {k: (ra[k]["avg_launch_ms"], ra[k]["frac"]) for k in ("sweep_C", "sweep_EA", "advect_stage") if k in ra}, d["verified"]["ok"])
PY
cp profiles/r04_*.json profiles/r04_*.txt $OUT/ 2>/dev/null
ls $OUT | grep r04 | head -30
