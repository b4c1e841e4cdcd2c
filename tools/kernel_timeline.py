#!/usr/bin/env python3
"""One BiCGSTAB iteration as the GPU saw it, from a rocprofv3 --kernel-trace CSV: every kernel (any stream) between two
consecutive starts of the anchor kernel, with its start relative to the anchor, duration and the idle time before it.
usage: python tools/kernel_timeline.py <..._kernel_trace.csv> [anchor substring, default k_edge<3] [which occurrence, default 20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_edge<3"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
short = lambda n: n.split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:44]
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) <= which + 1:
    sys.exit("only %d launches of %s" % (len(idx), anchor))
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print("one iteration (%s #%d -> #%d): %.1f us" % (anchor, which, which + 1, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%7.1f us  %6.1f us  (idle before: %5.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(r["Kernel_Name"])))
    prev_end = max(prev_end, e)
# the same over many iterations: median period
per = sorted(int(rows[j]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i, j in zip(idx[5:], idx[6:]))
print("median period over %d iterations: %.1f us" % (len(per), per[len(per) // 2] / 1e3))
