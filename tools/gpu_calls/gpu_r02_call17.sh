#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $1"; shift; env "$@" VARIANTS=fused1 SKIP_CHECK=1 timeout 300 python tools/gpu_variants.py 2>&1 | grep -E "TIME|Error|error" | sed 's/finish_in_kernel=1 n=4096 //; s/advect_stage.*A=/A=/; s/scalars.*//; s/iters=50 err=[^ ]* //'; }
run "full" X=1
run "no ring loads/jobs (dbg 1)" CUP2D_FUSED_DBG=1
run "no MFMA at all (dbg 6)" CUP2D_FUSED_DBG=6
run "no ring, no MFMA (dbg 7)" CUP2D_FUSED_DBG=7
run "ring without its MFMA (dbg 2)" CUP2D_FUSED_DBG=2
run "full" X=1
