#!/bin/bash
# edge form with the three-buffer load schedule: parity + timing
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 tools/gpu_edge_check.py check time > $OUT/r03_edge_check2.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import json
for line in open("gpurun_out/r03_edge_check2.log"):
    p = line.split(" ", 2)
    if p[0] == "check":
        try:
            d = json.loads(p[2]); print("check", p[1], {k: (v["4"]["rel_diff_last_iterate"], v["50"]["err"]) for k, v in d.items()} if "hilbert 8x8" in d else d)
        except Exception as e:
            print(line[:600])
    else:
        print(line.strip()[:600])
PY
