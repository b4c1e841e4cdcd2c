#!/bin/bash
# round 5, call 9: the overlap organisation (split sweeps + deferred update: the block transfer behind the inner launch, one
# all-gather exposed per reduction point) -- parity (= round 4's split sweeps bit for bit), step times next to the default on the
# four-sided patches, its timeline
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_comm.py -q -m gpu -p no:cacheprovider -k "received_in_place" > $OUT/r05c9_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/r05c9_pytest.log | tail -5
grep -n "Error\|assert \|Traceback" $OUT/r05c9_pytest.log | cut -c1-400 | head -20
for nby in 512 256; do
  for org in "1,0" "1,1" "0,1"; do
    NBY=$nby AXES=xy STEPS=6 ORG=$org timeout 300 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -E "ms/step|N-rank path" | cut -c1-160
  done
done
cd /tmp; rm -rf /tmp/prof_ov
AXES=xy STEPS=3 ORG=1,1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r05c9_self_ov.log 2>&1
f=$(find /tmp/prof_ov -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 4" 40 | tee $GRAFT_REPO_ROOT/$OUT/r05_nrank_timeline_overlap.txt
