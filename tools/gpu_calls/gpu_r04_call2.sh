#!/bin/bash
# round 4, call 2: prolongation / restriction kernels (cup2d_amr_regrid_device) -- fuzz vs host regrid, adapt() by every route vs the
# reference's adapt(), the C++ driver -- and the stage breakdown of a regrid on the 63 k-block grid
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_amr.py tests/test_distributed.py -q -m gpu -p no:cacheprovider -s \
  -k "regrid_kernels or adapt or cpp_host_driver or driver_amr or run_with_regridding" > $OUT/r04c2_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "passed|failed|FAILED|Error|assert|reference adapt" $OUT/r04c2_pytest.log | tail -30
python3 - <<'PY' 2>&1 | grep -v "^\[cup2d\] beat" | tail -80
import sys, json, argparse, time
sys.path.insert(0, ".")
import bench
a = argparse.Namespace(amr_lfine=9, iters=50, math="fast")
bench.beat = lambda *x: None
import os
for rep in range(3):
    if rep == 2: os.environ["CUP2D_HOST_TIMING"] = "1"
    r = bench.amr_leg(a, 0)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "operator_install_ms", "regrid")}))
PY
