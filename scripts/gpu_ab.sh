#!/bin/bash
# A/B of kernel variants: runs bench.py once per library under build_variants/ (PLP_LIB override).
cd "${GRAFT_REPO_ROOT:-.}"
for lib in build_variants/*.so; do
  echo "== $lib"
  PLP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   LP/s %.4g  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
