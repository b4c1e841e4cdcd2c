#!/bin/bash
# round 6, call 59: after the last change to tune_placement: solver tests, N-rank tests, world-8 bench test
set -u
export TMPDIR=/tmp
timeout 2000 python3 -m pytest tests/test_solver_variants_gpu.py tests/test_baseline_sizes_gpu.py tests/test_bench_world8.py tests/test_comm.py tests/test_distributed.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
