#!/bin/bash
# round-2 GPU call 1: full -m gpu suite (incl. the BASELINE-size parity tests), bench (single + one-rank communicator
# path), fused-kernel variants.  Everything lands under gpurun_out/.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt; rocm-smi --showmeminfo vram 2>/dev/null | head -8 >> $OUT/host.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 -s > $OUT/r02_pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/r02_pytest_gpu.log
grep -E "passed|failed|error|bench-config|4096\^2 step|FAILED|Error" $OUT/r02_pytest_gpu.log | tail -40
timeout 600 python bench.py > $OUT/r02_bench.json 2> $OUT/r02_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02_bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "verified")}, {k: (v["avg_launch_ms"], v["frac"]) for k, v in d["roofline_all"].items()}, d["cpu_baseline"])
except Exception as e:
    print("bench parse", e)
PY
timeout 600 python bench.py --force-dist --no-cpu-baseline > $OUT/r02_bench_fd.json 2> $OUT/r02_bench_fd.err; echo "bench force-dist rc=$?"; tail -3 $OUT/r02_bench_fd.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02_bench_fd.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "verified")}, d["config"]["comm"])
except Exception as e:
    print("bench fd parse", e)
PY
for v in "" cup2d_amd/variants/libcup2d_hip_0xED9_w4d2.so cup2d_amd/variants/libcup2d_hip_0xED9_w4d3.so cup2d_amd/variants/libcup2d_hip_0xED9_w4d3p.so; do
  echo "== variant ${v:-default}"
  CUP2D_LIB=${v:+$PWD/$v} VARIANTS=fused1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "CHECK|TIME|VARIANTS|Error|error" | tail -6
done 2>&1 | tee $OUT/r02_variants1.log
echo "total $(( $(date +%s) - t0 )) s"
