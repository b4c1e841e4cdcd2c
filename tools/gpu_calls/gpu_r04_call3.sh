#!/bin/bash
# round 4, call 3: the two timing modes: how do fresh processes on ONE box distribute, and does a start
# offset of the large buffers below the allocation granularity move them?  Each line: skew, ms/step, C+D' us, E+A+B us
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # $1 = label, rest = env
  local label=$1; shift
  for rep in 1 2 3; do
    env "$@" python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-amr --no-verify > $OUT/r04c3_tmp.json 2> $OUT/r04c3_tmp.err
    python3 - "$label" "$rep" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r04c3_tmp.json").read().strip().splitlines()[-1])
    k = d["kernels"]
    print("%-28s rep %s: %.3f ms/step  C+D' %.1f us  E+A+B %.1f us  advect %.1f us" % (sys.argv[1], sys.argv[2], d["ms_per_step"],
          1e3 * k["sweep_C"]["ms_avg"], 1e3 * k["sweep_EA"]["ms_avg"], 1e3 * k["advect_stage"]["ms_avg"]), flush=True)
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/r04c3_tmp.err").read()[-300:])
PY
  done
}
run "default" CUP2D_DUMMY=1
run "skew 4352 (4K+256)" CUP2D_ALLOC_SKEW=4352
run "skew 69888 (64K+4K+256)" CUP2D_ALLOC_SKEW=69888
run "skew 1052928 (1M+4K+256)" CUP2D_ALLOC_SKEW=1052928
run "skew 2101504 (2M+4K+256)" CUP2D_ALLOC_SKEW=2101504
run "pool off" CUP2D_POOL=0
run "default again" CUP2D_DUMMY=1
