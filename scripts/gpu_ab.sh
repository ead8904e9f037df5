#!/bin/bash
# A/B of kernel variants ON ONE BOX (boxes differ by +-2 %): bench.py once per library under build_variants/ (PLP_LIB
# override) and once with the in-tree library, two rounds.   gpurun --timeout 900 -- 'bash scripts/gpu_ab.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step %.4f  kernel_ms %.4f  LP/s %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"; }
for r in 1 2; do
  for lib in build_variants/*.so; do
    echo -n "$(basename $lib)"; PLP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | show
  done
  echo -n "in-tree"; timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | show
done
