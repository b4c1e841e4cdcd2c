#!/bin/bash
# The round's gate, exactly as the driver runs it: the GPU suite (driver's command line), smoke(), the bench line.
# Run through gpurun AFTER the last kernel commit of a round:   gpurun --timeout 1500 -- bash tools/gpu_final_gate.sh
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/gate_pytest.log 2>&1   # (the driver's command line, verbatim)
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -5 $OUT/gate_pytest.log
grep -c "^\[cup2d\] start" $OUT/gate_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/gate_bench.json 2> $OUT/gate_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/gate_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step") if k in d}, d.get("roofline"), d.get("verified"))
except Exception as e:
    print("bench line unreadable:", e)
PY
# what kind of box this was (the two timing modes are per box: DESIGN.md section 6)
rocm-smi --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition 2>&1 | grep -E "fclk|mclk|sclk|Power|Partition" | head -8
