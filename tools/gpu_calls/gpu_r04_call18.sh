#!/bin/bash
# round 4, call 18: do the two timing modes follow the placement of the vectors?  ten contexts alive in one process, twice
set -u
export TMPDIR=/tmp
python3 tools/gpu_placement_modes.py 2>&1 | tail -14
echo ---- a second process
python3 tools/gpu_placement_modes.py 2>&1 | tail -14
