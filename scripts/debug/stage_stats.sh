#!/bin/bash
# Debug build of the fused-reduce TU with -DPLP_STAGE_STATS (per-stage iteration counters, plp_reduce_r_impl.hpp) linked
# with the regular objects into build_variants/libplp_hip_stats.so (travels to the GPU box), then:
#   gpurun -- 'python scripts/debug/stage_stats.py'
set -e
cd "$(dirname "$0")/../../polytope_amd/csrc"
make -j16 >/dev/null
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DPLP_STAGE_STATS \
    -c plp_reduce_r.hip -o /tmp/plp_reduce_r_stats.o
objs=$(ls *.o | grep -v '^plp_reduce_r\.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libplp_hip_stats.so $objs /tmp/plp_reduce_r_stats.o -Wl,-rpath,/opt/rocm/lib
echo built build_variants/libplp_hip_stats.so
