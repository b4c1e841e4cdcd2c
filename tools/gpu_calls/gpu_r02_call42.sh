#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 1 ]; then export CUP2D_EXPERIMENT_BLIND=1; fi
  cd $R; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-amr --no-kernel-timers --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blind=$v', d['value'], d['ms_per_step'])"
  cd /tmp && rm -rf /tmp/gap_$v && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_$v -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-amr --no-kernel-timers --no-verify > /dev/null 2>&1
  python $R/tools/kernel_gaps.py $(find /tmp/gap_$v -name "*kernel_trace.csv" | head -1) | sed -n 3p
done
