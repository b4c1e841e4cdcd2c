#!/bin/bash
# L2-miss traffic of the fused sweeps: full form, edge form, edge form with sibling sharing (FETCH_SIZE / WRITE_SIZE passes)
set -u
export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-amr --no-verify"
echo "== full"; bash tools/gpu_quick_traffic.sh full $CMD 2>&1 | grep -E "rc=|k_fused|k_edge|k_sweepE"
echo "== edge"; CUP2D_FUSED_FORM=edge CUP2D_EDGE_SHARE=0 bash tools/gpu_quick_traffic.sh edge $CMD 2>&1 | grep -E "rc=|k_fused|k_edge|k_sweepE"
echo "== edge+share"; CUP2D_FUSED_FORM=edge bash tools/gpu_quick_traffic.sh edgeshare $CMD 2>&1 | grep -E "rc=|k_fused|k_edge|k_sweepE"
