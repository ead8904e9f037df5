#!/bin/bash
# A build of the library with ONE source recompiled under extra flags, the other objects the in-tree ones:
#   scripts/debug/one_file_variant.sh <name> <file.hip> -DPLP_X=1 ...   ->  build_variants/<name>.so   (run with PLP_LIB=...)
set -e
NAME=$1; SRC=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/polytope_amd/csrc"
mkdir -p "$ROOT/build_variants"
OBJ=${SRC%.hip}.o
OBJS=$(ls *.o | grep -v "^$OBJ$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -I../../include -c "$SRC" -o /tmp/plp_var_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/$NAME.so" $OBJS /tmp/plp_var_$NAME.o -Wl,-rpath,/opt/rocm/lib
echo "built build_variants/$NAME.so ($SRC $*)"
