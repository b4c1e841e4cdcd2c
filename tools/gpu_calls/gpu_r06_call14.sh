#!/bin/bash
# round 6, call 14: what distinguishes a box without a fast placement: socket power and clocks sampled while the step runs; on such
# a box also the hand-over masks of the edge-form sweeps
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.3; done ) > /tmp/smi.log 2>&1 &
SMI=$!
CUP2D_EDGE_SHARE=5 timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | tee /tmp/first.log
kill $SMI 2>/dev/null
python3 - <<'PY'
import re
pw, sc, mc = [], [], []
for l in open("/tmp/smi.log"):
    m = re.search(r"Power \(W\): ([0-9.]+)", l)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: sc.append(int(m.group(1)))
    m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: mc.append(int(m.group(1)))
if pw: print("socket power while the step runs: max %.0f W, mean of the upper half %.0f W (%d samples); sclk levels seen %s; mclk %s" % (max(pw), sum(sorted(pw)[len(pw)//2:]) / max(1, len(pw) - len(pw)//2), len(pw), sorted(set(sc)), sorted(set(mc))))
PY
if grep -q "placement kept 3[5-9][0-9]" /tmp/first.log; then
  echo "== a box without a fast placement: hand-over masks"
  for SH in 15 13 7 5 15; do CUP2D_EDGE_SHARE=$SH timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1; done
  timeout 200 python3 tools/gpu_advect_stages.py 4096 8 2>&1 | tail -1
fi
