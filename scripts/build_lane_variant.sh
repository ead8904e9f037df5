#!/bin/bash
# Variant of libplp_hip.so that differs from the in-tree build ONLY in plp_reduce_lane.hip (extra -D flags): that one file
# is compiled again and linked with the in-tree objects (seconds instead of the minutes of scripts/build_variant.sh).
#   scripts/build_lane_variant.sh <name> -DPLP_X=1 ...   ->  build_variants/<name>.so   (A/B on one box: PLP_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/polytope_amd/csrc
mkdir -p "$ROOT/build_variants"
O=/tmp/plp_lane_variant_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$C/plp_reduce_lane.hip" -o "$O" 2>&1 | grep -E "Name: _ZN3plp18reduce_lane_kernelILi3|VGPRs:|ScratchSize" | tail -3
OBJS=$(ls "$C"/*.o | grep -v plp_reduce_lane.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/$NAME.so" $OBJS "$O" -Wl,-rpath,/opt/rocm/lib
echo "built build_variants/$NAME.so ($*)"
