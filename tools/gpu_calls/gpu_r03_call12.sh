#!/bin/bash
# knock-out builds of the edge form (timing only, wrong results): what a tile's time consists of
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=cup2d_amd/variants/libcup2d_hip_0xED9
E=CUP2D_FUSED_FORM=edge,CUP2D_EDGE_SHARE=0,SKIP_REL4=1
timeout 800 python3 tools/gpu_lib_variants.py default@$E ${V}_k1.so@$E ${V}_k2.so@$E ${V}_k3.so@$E ${V}_k7.so@$E ${V}_k15.so@$E ${V}_k8.so@$E default@$E 2>&1 | tee $OUT/r03_edge_knock.txt
