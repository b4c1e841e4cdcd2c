#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_solver_variants_gpu.py tests/test_abi_and_host.py -m gpu -q -x 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_distributed.py tests/test_baseline_sizes_gpu.py -m gpu -q -x -k "decomposed or whole_step or cpp_mpi" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-amr > $OUT/r02_bench39.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench39.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_no_kernel_timers"], d["verified"]["ok"])
for name,v in d["kernels"].items():
    if v.get("launches"): print("  %-16s %8.1f us x %d"%(name, 1e3*v["ms_total"]/v["launches"], v["launches"]))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-amr --no-kernel-timers --no-verify > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/fr/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=5
tot=0
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"])):
    n=r["Name"]
    if any(k in n for k in ("k_fused","k_sweepE","k_smoother")): continue
    per=float(r["TotalDurationNs"])/steps/1e3
    tot+=per
    if per>3: print("   %-60s %7.1f us/step (%s calls)"%(n[:60],per,r["Calls"]))
print("fringe per step: %.0f us"%tot)
PY
