#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by tools/gpu_profile.sh under gpurun_out/ into the tracked summaries
under profiles/:
   profiles/<tag>_kernel_stats.txt   per-kernel calls / total / avg / min / max (rocprofv3 --kernel-trace --stats)
   profiles/<tag>_pmc_traffic.json   per-kernel HBM bytes per launch from the FETCH_SIZE and WRITE_SIZE passes
FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read at
exactly half its bytes (MI355X_MICROARCH.md, HBM): it is doubled here.  The correction is checked on every
run against kernels whose read volume is known exactly (k_reduce_partial reads one field, k_shift<0> one
scalar field) -- see "calibration" in the JSON."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def short(name):
    name = name.replace("cup2d::", "").replace("void ", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


stats_csv = os.path.join(ROOT, out_dir, "prof_%s" % tag, "stats_kernel_stats.csv")
if os.path.exists(stats_csv):
    rows = list(csv.DictReader(open(stats_csv)))
    full = {}
    full_csv = os.path.join(os.path.dirname(stats_csv), "stats_full_launches.csv")
    if os.path.exists(full_csv):
        full = {r["Name"]: r for r in csv.DictReader(open(full_csv))}
    cmd = "python bench.py --steps %s --warmup 1 --no-cpu-baseline --no-amr" % os.environ.get("STEPS", "2")
    with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- %s   (MI355X, gfx950)\n" % cmd)
        f.write("# source: %s (stats_kernel_stats.csv), durations in ns\n" % out_dir)
        f.write("# WorkCalls / WorkAvg(ns): launches that did work -- a BiCGSTAB sweep enqueued behind a finished solve returns at\n"
                "# once (a few us, see Min); bench.py's avg_launch_ms is comparable to WorkAvg, not to Avg\n")
        f.write("%-58s %7s %14s %12s %10s %10s %7s %9s %12s\n" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct",
                                                                  "WorkCalls", "WorkAvg(ns)"))
        for r in rows:
            fl = full.get(r["Name"])
            f.write("%-58s %7s %14s %12.0f %10s %10s %6.2f%% %9s %12s\n" % (short(r["Name"])[:58], r["Calls"], r["TotalDurationNs"],
                                                                          float(r["AverageNs"]), r["MinNs"], r["MaxNs"], float(r["Percentage"]),
                                                                          fl["FullCalls"] if fl else "-",
                                                                          "%.0f" % float(fl["FullAvgNs"]) if fl else "-"))
    print("wrote profiles/%s_kernel_stats.txt" % tag)

traffic = {}
for sub, ctr in (("pmc_fetch_%s" % tag, "FETCH_SIZE"), ("pmc_write_%s" % tag, "WRITE_SIZE")):
    path = os.path.join(ROOT, out_dir, sub, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # launches that return at once (speculative BiCGSTAB iterations enqueued behind a finished solve) move no
        # data: they are not part of the per-launch average
        full = [x for x in v if x >= 0.05 * max(v)] or v
        traffic.setdefault(k, {})[ctr] = {"launches": len(full), "early_exits": len(v) - len(full),
                                          "avg_KiB": sum(full) / len(full), "min_KiB": min(full), "max_KiB": max(full)}
if traffic:
    outj = {"unit": "bytes per launch", "correction": "hbm_bytes = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024 (gfx950 FETCH_SIZE x2)",
            "kernels": {}}
    for k, d in sorted(traffic.items()):
        fe = d.get("FETCH_SIZE", {}).get("avg_KiB")
        wr = d.get("WRITE_SIZE", {}).get("avg_KiB")
        e = {"fetch_KiB_raw": fe, "write_KiB_raw": wr, "launches": d.get("FETCH_SIZE", d.get("WRITE_SIZE"))["launches"]}
        if fe is not None and wr is not None:
            e["read_bytes"] = 2.0 * fe * 1024
            e["write_bytes"] = wr * 1024
            e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
        outj["kernels"][k] = e
    json.dump(outj, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag), "w"), indent=1, sort_keys=True)
    print("wrote profiles/%s_pmc_traffic.json" % tag)
