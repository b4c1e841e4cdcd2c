#!/bin/bash
# round 4, call 20: is the mode a property of single buffers or of their combination?
export TMPDIR=/tmp
./tools/placement_probe.bin 44 2>&1 | tail -60
