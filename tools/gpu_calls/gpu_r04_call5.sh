#!/bin/bash
# round 4, call 5: the N-rank code path of a whole step on one GPU, bytes through RCCL (self-periodic patch) vs the plain context
set -u
export TMPDIR=/tmp
python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -v "^\[" | tail -12
NBX=512 NBY=256 python3 tools/gpu_selfperiodic_step.py 2>&1 | grep -v "^\[" | tail -8
