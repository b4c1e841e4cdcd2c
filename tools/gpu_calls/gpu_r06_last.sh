#!/bin/bash
# round 6: the GPU suite and smoke on the round's last commit (the bench line: gpu_r06_call50.sh)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1800 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/last_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/last_pytest.log | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
