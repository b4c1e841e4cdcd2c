#!/bin/bash
# round 5, call 8 (records, no new code): the size sweep, the timeline of configs[3]'s per-rank patch with four ghost sides, and
# `bench.py --gpus 8` at FULL size with the eight ranks sharing the GPU (not a measurement: that every line of an 8-rank run of
# both layouts at their real sizes has executed)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SIDE="--no-cpu-baseline --no-amr --no-nrank-proxy --no-second-size --no-north-star-floors --no-tolerance-leg"
rm -f $OUT/r05_size_sweep.jsonl
for n in 2048 4096 8192; do
  timeout 600 python3 bench.py --gpus 1 --n $n --steps 10 --warmup 3 $SIDE 2>/dev/null | tail -1 >> $OUT/r05_size_sweep.jsonl
done
timeout 600 python3 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 $SIDE 2>/dev/null | tail -1 >> $OUT/r05_size_sweep.jsonl
python3 - <<'PY'
import json
out = []
for line in open("gpurun_out/r05_size_sweep.jsonl"):
    d = json.loads(line)
    ra = d["roofline_all"]
    row = {"workload": d["config"]["workload"][:24], "parallelism": d["config"]["parallelism"], "value": d["value"], "ms_per_step": d["ms_per_step"],
           "kernels_ms_frac": {k: (ra[k]["avg_launch_ms"], ra[k]["frac"]) for k in ("sweep_C", "sweep_EA", "advect_stage") if k in ra},
           "verified_ok": d["verified"]["ok"], "placement_kept_slowest_us": [round((d.get("placement") or {}).get(k, 0), 1) for k in ("kept_us", "slowest_us")]}
    out.append(row)
    print(row)
json.dump(out, open("gpurun_out/r05_size_sweep.json", "w"), indent=1)
PY
cd /tmp; rm -rf /tmp/prof_c3
NBY=256 AXES=xy STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r05c8_self_c3.log 2>&1
grep -v "^W2026\|^E2026\|simple_timer" $GRAFT_REPO_ROOT/$OUT/r05c8_self_c3.log | tail -7 | cut -c1-500
f=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r05_nrank_timeline_configs3.txt
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
CUP2D_BENCH_SHARE_GPU=1 CUP2D_BENCH_WATCHDOG_S=280 OMP_NUM_THREADS=4 timeout 900 python3 bench.py --gpus 8 --steps 3 --warmup 1 --layout configs3 --no-cpu-baseline \
  > $OUT/r05_bench_world8_shared_gpu.json 2> $OUT/r05_bench_world8_shared_gpu.err
echo "bench world 8 (shared GPU) rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_world8_shared_gpu.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"]["global_grid"], d["config"]["parallelism"], (d.get("verified") or {}).get("ok"))
    print("second_layout", d.get("second_layout"))
    print("comm", json.dumps(d["config"]["comm"])[:900])
except Exception as e:
    print("unreadable:", e)
PY
tail -4 $OUT/r05_bench_world8_shared_gpu.err | cut -c1-300
