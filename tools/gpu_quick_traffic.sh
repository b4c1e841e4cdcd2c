#!/bin/bash
# tools/gpu_quick_traffic.sh <tag> <command...> -- FETCH_SIZE / WRITE_SIZE per kernel of any command (two PMC passes), summary on stdout
set -u
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/qf_$TAG $OUT/qw_$TAG
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/qf_$TAG -o pmc -- "$@" > $OUT/qf_$TAG.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/qw_$TAG -o pmc -- "$@" > $OUT/qw_$TAG.log 2>&1; echo "write rc=$?"
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("qf", "qw"):
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (d, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void cup2d::", "").replace("cup2d::", "")[:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, a in acc.items():
    if not k.startswith("k_"): continue
    rd = 2 * 1024 * sum(a["FETCH_SIZE"]) / max(1, len(a["FETCH_SIZE"])); wr = 1024 * sum(a["WRITE_SIZE"]) / max(1, len(a["WRITE_SIZE"]))
    print("%-34s launches %3d  read %7.1f MB  write %7.1f MB  total %7.1f MB (FETCH_SIZE x2, gfx950)" % (k, len(a["FETCH_SIZE"]), rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
PY
