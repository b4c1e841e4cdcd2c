#!/bin/bash
# round 6, call 43: the export to the siblings in front of the next tile's requests (A/B against the previous order), then the phases again
set -u
export TMPDIR=/tmp
V=cup2d_amd/variants
for L in $V/libcup2d_hip_0xED9_old.so "" $V/libcup2d_hip_0xED9_old.so ""; do
  echo "lib ${L:-new}: $(CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
for L in $V/libcup2d_hip_0xED9_old.so "" $V/libcup2d_hip_0xED9_old.so ""; do
  echo "lib ${L:-new}: $(N=2048 CUP2D_LIB=$L timeout 200 python3 tools/gpu_share_ab.py 2>&1 | tail -1 | cut -c1-200)"
done
CUP2D_LIB=$V/libcup2d_hip_0xED9_ph.so timeout 300 python3 tools/gpu_share_ab.py 2>&1 | grep -E "EPHASES" | python3 -c "
import sys, re, collections
acc = collections.defaultdict(lambda: [0, None])
for l in sys.stdin:
    m = re.match(r'EPHASES mode (\d) wg \d wave (\d) tiles (\d+) cycles/tile: (.*)', l)
    if not m: continue
    vals = [int(x) for x in re.findall(r' (\d+)(?=  |\$)', ' ' + m.group(4) + '  ')]
    names = re.findall(r'([a-z+ ()]+?) \d+', m.group(4))
    k = (m.group(1), m.group(2))
    a = acc[k]
    a[0] += 1
    a[1] = vals if a[1] is None else [x + y for x, y in zip(a[1], vals)]
    acc[k] = a
for k, (n, v) in sorted(acc.items()):
    print('mode %s wave %s  (%d samples)  per tile:' % (k[0], k[1], n), [round(x / n) for x in v], 'sum', round(sum(v) / n))
"
