#!/bin/bash
# round 6, call 37: which resource differs between a slow and a fast placement of the eleven vectors (option B):
# the same three steps on the context's own set (CUP2D_PLACEMENT_TRIES=1: a slow one on these boxes) and on the searched + repaired set,
# four counter passes each (separate --pmc runs, no trace flags)
set -u
export TMPDIR=/tmp
cd /tmp
P1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P2="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
P3="TCC_TAG_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_UTCL2_BUSY"
P4="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
for T in 1 8; do
  for K in 1 2 3 4; do
    eval PM=\$P$K
    rm -rf /tmp/pc
    CUP2D_PLACEMENT_TRIES=$T timeout 300 rocprofv3 --pmc $PM --output-format csv -d /tmp/pc -o c -- python3 $GRAFT_REPO_ROOT/tools/gpu_placement_counters.py > /tmp/pc.log 2>&1
    echo "tries $T pass $K rc=$? $(grep placement /tmp/pc.log | cut -c1-160)"
    f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
    python3 - "$f" $T <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_edge<2, 1," in k or "k_edge<3, 1," in k:
        kk = "E+A+B" if "k_edge<2, 1," in k else "C+D'"
        a = acc[(kk, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (kk, cn), (v, n) in sorted(acc.items()):
    print("   tries %s  %-6s %-40s %16.0f per launch (%d launches)" % (sys.argv[2], kk, cn, v / n, n))
PY
  done
done
