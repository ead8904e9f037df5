#!/bin/bash
# Build a variant of libplp_hip.so with extra -D flags into build_variants/<name>.so (for same-box A/B runs):
#   scripts/build_variant.sh <name> -DPLP_X=1 ...      (the in-tree library and objects are left alone)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/plp_variant_$NAME
rm -rf "$W"; mkdir -p "$W/polytope_amd/csrc" "$W/include" "$ROOT/build_variants"
cp "$ROOT"/polytope_amd/csrc/*.hip "$ROOT"/polytope_amd/csrc/*.hpp "$ROOT"/polytope_amd/csrc/Makefile "$W/polytope_amd/csrc/"
cp "$ROOT"/include/*.h "$W/include/"
make -s -C "$W/polytope_amd/csrc" -j"$(nproc)" HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $*"
cp "$W/polytope_amd/libplp_hip.so" "$ROOT/build_variants/$NAME.so"
rm -rf "$W"
echo "built build_variants/$NAME.so ($*)"
