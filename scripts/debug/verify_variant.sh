#!/bin/bash
# A build of the library with plp_verify.hip alone recompiled under extra flags, the other objects the in-tree ones:
#   scripts/debug/verify_variant.sh <name> -DPLP_VF_LPB17=4 ...   ->  build_variants/<name>.so   (run with PLP_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/polytope_amd/csrc"
mkdir -p "$ROOT/build_variants"
OBJS=$(ls *.o | grep -v plp_verify.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -I../../include -c plp_verify.hip -o /tmp/plp_verify_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/$NAME.so" $OBJS /tmp/plp_verify_$NAME.o -Wl,-rpath,/opt/rocm/lib
echo "built build_variants/$NAME.so ($*)"
