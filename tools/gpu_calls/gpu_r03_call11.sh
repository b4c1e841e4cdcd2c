#!/bin/bash
# the edge form with the job-ahead load schedule (no spills) against the default form: check of the last iterates, then timers
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python3 tools/gpu_edge_check.py > $OUT/r03_edge_deep_check.txt 2>&1; tail -25 $OUT/r03_edge_deep_check.txt
REPS=2 timeout 500 python3 tools/gpu_lib_variants.py default default@CUP2D_FUSED_FORM=edge default@CUP2D_FUSED_FORM=edge,CUP2D_EDGE_SHARE=0 2>&1 | tee $OUT/r03_edge_deep_ab.txt
