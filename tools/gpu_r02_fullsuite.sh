#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/r02_pytest_full.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -16 $OUT/r02_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
