#!/bin/bash
# cache policy of the streams MODE 2 / 3 add (CUP2D_POLICY2)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip_0xED9_p2
E=SKIP_REL4=1
REPS=2 timeout 800 python3 tools/gpu_lib_variants.py default@$E ${V}_0x01.so@$E ${V}_0x07.so@$E ${V}_0x08.so@$E ${V}_0x30.so@$E ${V}_0xC0.so@$E 2>&1 | tee $OUT/r03_eab_policy2.txt
