#!/bin/bash
# FAST advect through the plan's quads + leftovers on odd grids / other block orders; FAST on N ranks sharing the GPU (phases, ghost blocks)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fast_advect_on_quads" 2>&1 | tail -3
timeout 400 python -m pytest tests/test_distributed.py -m gpu -q -k "decomposed_step" 2>&1 | tail -3
