#!/bin/bash
# round 5, call 3: the deferred N-rank organisation (records in the send/recv group, MERGE 3) -- parity with the other paths, the
# 2 x 4 tests with their corrected tolerances, the per-iteration timeline of the doubly periodic patch
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python3 -m pytest tests/test_comm.py tests/test_distributed.py -q -m gpu -p no:cacheprovider -s \
  -k "periodic or received_in_place or one_rank or 8-2-4 or two_by_four or 2-2-1-8-16" > $OUT/r05c3_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "^FAILED|^ERROR|passed|failed|gpu_big" $OUT/r05c3_pytest.log | tail -12
grep -n "first worker traceback" -A 25 $OUT/r05c3_pytest.log | cut -c1-300 | head -60
grep -n "error lines" -A 6 $OUT/r05c3_pytest.log | cut -c1-400 | head -30
grep -n "Error\|assert" $OUT/r05c3_pytest.log | cut -c1-300 | head -20
cd /tmp
for ax in xy; do
  rm -rf /tmp/prof_$ax
  AXES=$ax STEPS=3 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$ax -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r05c3_self_$ax.log 2>&1
  tail -8 $GRAFT_REPO_ROOT/$OUT/r05c3_self_$ax.log | cut -c1-600
  f=$(find /tmp/prof_$ax -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, 3" 40 | tee $GRAFT_REPO_ROOT/$OUT/r05_nrank_timeline.txt
done
NBY=256 AXES=xy STEPS=5 timeout 300 python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py 2>&1 | tail -6 | cut -c1-600
