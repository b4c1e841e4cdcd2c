#!/usr/bin/env python3
"""A whole time step as the GPU saw it (rocprofv3 --kernel-trace CSV): the kernels between two consecutive starts of an anchor
kernel that runs once per step, iterations of the solver collapsed to their first and to a summary line; idle time per kind.
usage: python tools/kernel_step_timeline.py <..._kernel_trace.csv> <anchor substring> [which occurrence, default -2]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
short = lambda n: n.split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:46]
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
span = int(rows[b]["Start_Timestamp"]) - t0
busy, prev_end = 0, t0
gap_after = collections.defaultdict(lambda: [0, 0])
cnt = collections.Counter()
dur = collections.Counter()
first_seen = {}
for r in rows[a:b]:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    g = max(0, s - prev_end)
    busy += max(0, e - max(s, prev_end))
    gap_after[n][0] += g; gap_after[n][1] += 1
    cnt[n] += 1; dur[n] += e - s
    first_seen.setdefault(n, (s - t0) / 1e3)
    prev_end = max(prev_end, e)
print("step: %.1f us between two %s; GPU busy %.1f us, idle %.1f us (%d launches)" % (span / 1e3, anchor, busy / 1e3, (span - busy) / 1e3, b - a))
print("  %-46s %6s %10s %10s %12s %9s" % ("kernel", "calls", "total us", "avg us", "idle before", "first at"))
for n in sorted(cnt, key=lambda k: first_seen[k]):
    print("  %-46s %6d %10.1f %10.1f %12.1f %9.1f" % (n, cnt[n], dur[n] / 1e3, dur[n] / cnt[n] / 1e3, gap_after[n][0] / 1e3, first_seen[n]))
