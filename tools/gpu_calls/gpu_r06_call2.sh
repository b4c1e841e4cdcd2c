#!/bin/bash
# round 6, call 2: the pack folded into the deferred sweeps (k_edge MERGE 3 writes the send buffer, the record slot and r', p'' of
# the ghost blocks): the communicator tests, the per-iteration timeline with and without the fold, and the new bench line
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_comm.py tests/test_distributed.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
cd /tmp
for FOLD in 1 0; do
  for SHAPE in "512 512" "512 256"; do
    set -- $SHAPE
    rm -rf /tmp/prof_o
    CUP2D_FOLD_PACK=$FOLD NBX=$1 NBY=$2 AXES=xy STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r06c2_self_${FOLD}_$1x$2.log 2>&1
    echo "== fold $FOLD patch $1x$2"; grep -E "ms/step|N-rank path" $GRAFT_REPO_ROOT/$OUT/r06c2_self_${FOLD}_$1x$2.log | cut -c1-160
    f=$(find /tmp/prof_o -name "*kernel_trace.csv" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r06c2_timeline_${FOLD}_$1x$2.txt | head -16
  done
done
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/r06c2_bench.json 2> $OUT/r06c2_bench.err; echo "bench rc=$?"; tail -3 $OUT/r06c2_bench.err
wc -c $OUT/r06c2_bench.json; cat $OUT/r06c2_bench.json
cp gpurun_out/bench_detail.json $OUT/r06c2_bench_detail.json
