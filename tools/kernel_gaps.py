#!/usr/bin/env python3
"""Idle time between consecutive kernels of the solver loop, from a rocprofv3 --kernel-trace CSV.
usage: python tools/kernel_gaps.py <..._kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:30]
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na, nb = short(a["Kernel_Name"]), short(b["Kernel_Name"])
    if any(k in na for k in ("k_fused", "k_sweep", "k_hyb", "k_sell", "k_edge")) and int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) > 8000:
        gaps[(na, nb)].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:10]:
    v = sorted(v)
    print("  %-30s -> %-30s n=%5d  gap median %5.1f us  p90 %5.1f" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[int(len(v) * 0.9)] / 1e3))
