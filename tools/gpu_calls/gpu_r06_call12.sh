#!/bin/bash
# round 6, call 12: the host thread pool under the regrid (AMR tests on one and N ranks, the bench's regrid figure with stages)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_amr.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 1200 python3 -m pytest tests/test_distributed.py -m gpu -x -q -p no:cacheprovider -k "amr" 2>&1 | tail -2
CUP2D_HOST_TIMING=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nrank-proxy --no-second-size --no-tolerance-leg --no-north-star-floors > $OUT/r06c12_bench.json 2> $OUT/r06c12_bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r06c12_bench.json')); a=d['summary']['amr_configs4']; print(d['value'], a['value'], a['regrid_ms'], a['regrid_stages_ms'])
f=json.load(open('gpurun_out/bench_detail.json'))['amr_configs4']['regrid']; print([ (r['ms'], r['stages_ms']) for r in f['all']])"
grep "cup2d timing\] \(amr_install\|install_sell\)" $OUT/r06c12_bench.err | tail -6
