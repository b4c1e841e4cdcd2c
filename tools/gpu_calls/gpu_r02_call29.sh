#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_amr.py -m gpu -q -x -k "installed_from or tile_fused" > $OUT/r02_pytest29a.log 2>&1; echo "new rc=$?"; tail -3 $OUT/r02_pytest29a.log
timeout 600 python tools/gpu_amr_adapt_timing.py 2>&1 | cut -c1-150 | head -24
timeout 2400 python -m pytest tests/test_amr.py tests/test_distributed.py tests/test_comm.py -m gpu -q > $OUT/r02_pytest29b.log 2>&1; echo "amr+dist rc=$?"; tail -4 $OUT/r02_pytest29b.log
