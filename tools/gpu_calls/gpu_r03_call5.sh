#!/bin/bash
# new paths of the round on the GPU: N-rank AMR with per-rank regrid + migration, edge form of the fused sweeps, communicator self-test, bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_distributed.py tests/test_solver_variants_gpu.py tests/test_comm.py -x -q -m gpu -p no:cacheprovider -k "amr or edge or comm or rccl" > $OUT/r03_call5_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r03_call5_pytest.log
timeout 300 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-amr > $OUT/r03_call5_bench.json 2> $OUT/r03_call5_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_call5_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["verified"], indent=1)[:2500])
print({k: (v["frac"], v["avg_launch_ms"]) for k, v in d["roofline_all"].items()})
PY
