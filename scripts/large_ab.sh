#!/bin/bash
# Same-box A/B of the large-shape configurations: scripts/debug/large_ab.py once per library under build_variants/ and
# once with the in-tree library.   gpurun --timeout 900 -- 'bash scripts/large_ab.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in build_variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $lib)"; PLP_LIB=$PWD/$lib timeout 300 python scripts/debug/large_ab.py 2>&1 | grep -v "^{" | tail -20
done
echo "== in-tree"; timeout 300 python scripts/debug/large_ab.py 2>&1 | grep -v "^{" | tail -20
