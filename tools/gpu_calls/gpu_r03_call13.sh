#!/bin/bash
# first run of the organisation with sweep E and the next A+B in one launch (CUP2D_FUSED_FORM=eab): check, then timers
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
VARIANTS=eab+share,eab,full timeout 600 python3 tools/gpu_edge_check.py > $OUT/r03_eab_check.txt 2>&1; tail -12 $OUT/r03_eab_check.txt | cut -c1-2500
