#!/bin/bash
# round 4, call 25: a cell plan through real ncclSend / ncclRecv (rank = its own neighbour); the adapted-grid entry points of call 24
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_comm.py tests/test_amr.py -q -m gpu -p no:cacheprovider -k "cell_plan or rk_stages or without_an_installed or unsupported or fills_the_ghost" > $OUT/r04c25_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r04c25_pytest.log | tail -8
grep -E "^E  " $OUT/r04c25_pytest.log | head -30
