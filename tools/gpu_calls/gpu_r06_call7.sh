#!/bin/bash
# round 6, call 7: the operator of a regrid assembled by a work counter into a pinned staging buffer, uploaded asynchronously
# behind a parallel tiling (AMR tests, host stages, the bench's regrid figure); the tests this round touched
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_amr.py tests/test_comm.py tests/test_bench_world8.py -m gpu -x -q -p no:cacheprovider > $OUT/r06c7_tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/r06c7_tests.log | cut -c1-300
timeout 1500 python3 -m pytest tests/test_distributed.py -m gpu -x -q -p no:cacheprovider -k "16k or 4084 or amr_on_n_ranks" > $OUT/r06c7_tests_dist.log 2>&1; echo "dist tests rc=$?"; tail -4 $OUT/r06c7_tests_dist.log | cut -c1-300
CUP2D_HOST_TIMING=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nrank-proxy --no-second-size --no-tolerance-leg --no-north-star-floors > $OUT/r06c7_bench.json 2> $OUT/r06c7_bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r06c7_bench.json')); a=d['summary']['amr_configs4']; print(d['value'], a['value'], a['regrid_ms'], a['regrid_stages_ms'])"
grep "cup2d timing\] \(amr_install\|install_sell\)" $OUT/r06c7_bench.err | tail -12
