#!/bin/bash
# round 5, call 1: the 2 x 4 layout on 8 ranks sharing the GPU, the doubly self-periodic patch, bench.py at world 8, the bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_comm.py tests/test_bench_world8.py tests/test_distributed.py -q -m gpu -p no:cacheprovider -s \
  -k "periodic or fills_the_ghost or received_in_place or world_8 or 8-2-4 or two_by_four or 8-1 or 2-4-mpi" > $OUT/r05c1_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|gpu_big|cup2d_run_mpi" $OUT/r05c1_pytest.log | tail -12
t0=$(date +%s)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05c1_bench.json 2> $OUT/r05c1_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05c1_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "verified_summary") if k in d}, d.get("roofline"))
    print("placement", d.get("placement"))
    print("nrank", json.dumps(d.get("nrank_path_on_one_gpu"))[:1500])
except Exception as e:
    print("bench line unreadable:", e)
PY
