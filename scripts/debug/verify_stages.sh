#!/bin/bash
# Timing builds of the verifier that stop after stage k (k = 1..6) into build_variants/vf<k>.so: only plp_verify.hip is
# recompiled, the other objects are the in-tree ones.   scripts/debug/verify_stages.sh   (then, on the GPU box:
# for k in 1..6: PLP_LIB=build_variants/vf$k.so python scripts/debug/verify_smoke.py 0 time)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/polytope_amd/csrc"
mkdir -p "$ROOT/build_variants"
OBJS=$(ls *.o | grep -v plp_verify.o)
for k in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DPLP_VF_STOP=$k -I../../include -c plp_verify.hip -o /tmp/plp_verify_vf$k.o &
done
wait
for k in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/vf$k.so" $OBJS /tmp/plp_verify_vf$k.o -Wl,-rpath,/opt/rocm/lib
done
ls -la "$ROOT/build_variants/"
