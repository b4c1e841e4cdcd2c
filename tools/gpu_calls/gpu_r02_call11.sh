#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/gpu_amr_bench.py 2>&1 | tail -30
