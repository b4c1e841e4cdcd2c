#!/bin/bash
# CUP2D_FUSED_FORM=eab: descending C+D sweep (zigzag), sharing per kind of sweep
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
VARIANTS=eab,eab-noshare timeout 600 python3 tools/gpu_edge_check.py check > $OUT/r03_eab_check2.txt 2>&1; cut -c1-700 $OUT/r03_eab_check2.txt
VARIANTS=eab,eab-allshare,eab-noshare,eab-nozigzag,full timeout 600 python3 tools/gpu_edge_check.py time 2>&1 | tee $OUT/r03_eab_time2.txt
VARIANTS=eab,eab-nozigzag timeout 600 python3 tools/gpu_edge_check.py time 2>&1 | tee -a $OUT/r03_eab_time2.txt
