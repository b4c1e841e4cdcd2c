#!/bin/bash
# round 6, call 6: A/B of the projection's velocity store (non-temporal or not) in front of RK stage 1 on ONE box, with more
# conditions in between; the world-8 bench rehearsal with its stderr kept
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python3 tools/gpu_stage1_conditions.py 2>&1 | tail -8
CUP2D_LIB=$PWD/tools/ab/libcup2d_hip_projnt0.so timeout 300 python3 tools/gpu_stage1_conditions.py 2>&1 | tail -8
timeout 300 python3 tools/gpu_stage1_conditions.py 2>&1 | tail -8
CUP2D_BENCH_SHARE_GPU=1 CUP2D_BENCH_WATCHDOG_S=200 OMP_NUM_THREADS=4 timeout 800 python3 bench.py --gpus 8 --n 256 --steps 2 --warmup 1 --layout configs3 --configs3-n 1024 --iters 20 --no-cpu-baseline > $OUT/r06c6_w8.json 2> $OUT/r06c6_w8.err; echo "world8 rc=$?"
grep -v "^W0930\|^I0930\|^$" $OUT/r06c6_w8.err | grep -i -B5 -A25 "abort\|terminate\|Traceback\|Error" | head -120 | cut -c1-300
