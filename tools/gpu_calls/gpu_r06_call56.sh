#!/bin/bash
# round 6, call 56: where a tile's time goes in the sweeps of the adapted grid (k_fused HYB built with -DFUSED_PHASES)
set -u
export TMPDIR=/tmp
CUP2D_LIB=cup2d_amd/variants/libcup2d_hip_0xED9_fph.so LFINE=9 NOTIMING=1 timeout 300 python3 tools/gpu_amr_bench.py 2>&1 | grep -E "PHASES|AMR step" | python3 -c "
import sys, re, collections
acc = collections.defaultdict(lambda: [0, None])
for l in sys.stdin:
    if 'AMR step' in l: print(l.strip()[:120]); continue
    m = re.match(r'PHASES mode (\d) wg (\d+) wave (\d) tiles (\d+) cycles/tile: (.*)', l)
    if not m: continue
    vals = [int(x) for x in re.findall(r' (\d+)(?=  |\$)', ' ' + m.group(5).strip() + '  ')]
    k = (m.group(1), m.group(3), m.group(4))
    a = acc[k]; a[0] += 1
    a[1] = vals if a[1] is None else [x + y for x, y in zip(a[1], vals)]
for k, (n, v) in sorted(acc.items()):
    print('mode %s wave %s tiles/wave %s (%d samples) [ring stage+wait, ring mfma+edges, tile stage+wait, classify+issue, tile mfma, edges, stencil+stores]:' % (k[0], k[1], k[2], n), [round(x / n) for x in v], 'sum', round(sum(v) / n))
"
