#!/bin/bash
# round 6, call 4: the communicator test that failed in call 2 on the cleaned library (full output), what the launch in front
# of RK stage 1 does to it, the bench line with the floors measured in front of the product launch
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_comm.py -m gpu -x -q -p no:cacheprovider -k "ghost_blocks_received_in_place" > $OUT/r06c4_test_comm.log 2>&1; echo "test rc=$?"; tail -4 $OUT/r06c4_test_comm.log | cut -c1-300
grep -n "AssertionError\|assert \|Error" $OUT/r06c4_test_comm.log | head -20 | cut -c1-400
timeout 300 python3 tools/gpu_stage1_conditions.py 2>&1 | tail -6
python3 bench.py --steps 20 --warmup 5 > $OUT/r06c4_bench.json 2> $OUT/r06c4_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06c4_bench.json")); S = d["summary"]
print(d["value"], d["ms_per_step"], d["verified_ok"]); print(json.dumps(S["north_star"])); print(json.dumps(json.load(open("gpurun_out/bench_detail.json"))["roofline_north_star"].get("floors")))
PY
