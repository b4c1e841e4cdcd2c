#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for lib in cup2d_amd/variants/libcup2d_hip_oldrccl.so cup2d_amd/libcup2d_hip.so; do
  for m in "ours_first" "ours_first x" "torch_first" "torch_cuda_first"; do
    echo "=== $lib $m"; CUP2D_LIB=$PWD/$lib timeout 300 python tools/rccl_probe.py $m 2>&1 | grep -v "iommu\|^$" | tail -8
  done
done > $OUT/r02_probe22.log 2>&1
cat $OUT/r02_probe22.log
for k in "amr" "one_rank and True" "self"; do timeout 600 python -m pytest tests/test_comm.py -m gpu -q -k "$k" 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_comm.py -m gpu -q 2>&1 | tail -2
