#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 SKIP_CHECK=1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "TIME|Error|error" | sed 's/finish_in_kernel=1 n=4096 //; s/advect_stage.*A=/A=/; s/scalars.*//; s/iters=50 err=[^ ]* //'; }
run "E grid 2048 (default)" X=1
run "E grid 1024" CUP2D_GRID_E=1024
run "E grid 512" CUP2D_GRID_E=512
run "E grid 256" CUP2D_GRID_E=256
run "E grid 2048 (default)" X=1
