#!/bin/bash
# tools/build_policy_variants.sh <mask>... -- libcup2d_hip with krylov_fused.hip compiled under -DCUP2D_POLICY=<mask>
# (cache policy of the solver's streams; EXTRA_DEFS="-D..." adds defines, TAG names the variant), as
# cup2d_amd/variants/libcup2d_hip_<mask><TAG>.so; select one at run time
# with CUP2D_LIB=<path>.  Development aid.
set -eu
cd "$(dirname "$0")/../cup2d_amd/csrc"
make -s >/dev/null
mkdir -p ../variants
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DCUP2D_POLICY=$m ${EXTRA_DEFS:-} \
     -c krylov_fused.hip -o ../variants/krylov_fused_$m${TAG:-}.o &
done
wait
for m in "$@"; do
  objs=$(ls *.o | grep -v krylov_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcup2d_hip_$m${TAG:-}.so $objs ../variants/krylov_fused_$m${TAG:-}.o
  rm -f ../variants/krylov_fused_$m${TAG:-}.o
  echo built ../variants/libcup2d_hip_$m${TAG:-}.so
done
