#!/bin/bash
# round 6, call 1: FP64 op rates (v_rcp_f64 against fma: decides how the north-star kernel's reciprocals are batched), the
# round-5 library's bench line on this round's box, and the overlap organisation with CUs left free for the RCCL kernel
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
./tools/fp64_peak.bin 2>&1 | tee $OUT/r06c1_fp64_peak.txt
python3 bench.py --steps 20 --warmup 5 > $OUT/r06c1_bench.json 2> $OUT/r06c1_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06c1_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "verified", d["verified_ok"], "placement", d["placement"].get("kept_us"), d["placement"].get("slowest_us"))
print("north", {k: d["roofline_north_star"].get(k) for k in ("frac", "avg_launch_ms")}, d["roofline_north_star"].get("floors_on_this_box_us"))
print("second_size", d["second_size"], "amr", d["amr_configs4"].get("value"), "regrid", d["amr_configs4"].get("regrid", {}).get("ms"))
n = d["nrank_path_on_one_gpu"]; print("nrank", n.get("ratio_to_plain"), [o.get("ratio_to_plain") for o in n.get("other_patches", [])])
PY
cd /tmp
for SP in 0 8 16; do
  for ORG in 1,0 1,1; do
    rm -rf /tmp/prof_o
    CUP2D_SPARE_CUS=$SP ORG=$ORG AXES=xy STEPS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_selfperiodic_step.py > $GRAFT_REPO_ROOT/$OUT/r06c1_self_${SP}_${ORG}.log 2>&1
    echo "== spare $SP org $ORG"; grep -E "ms/step|N-rank path" $GRAFT_REPO_ROOT/$OUT/r06c1_self_${SP}_${ORG}.log | cut -c1-160
    f=$(find /tmp/prof_o -name "*kernel_trace.csv" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/kernel_timeline.py $f "k_edge<3, " 40 | tee $GRAFT_REPO_ROOT/$OUT/r06c1_timeline_${SP}_${ORG}.txt | head -24
  done
done
