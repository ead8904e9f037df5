#!/bin/bash
# reduce_wsplit_kernel against reduce_wdense_kernel on one box: same hashes = same bits.
cd "${GRAFT_REPO_ROOT:-.}"
for v in 0 2 4; do echo "== PLP_REDUCE_WSPLIT=$v"; PLP_REDUCE_WSPLIT=$v timeout 300 python scripts/debug/large_ab.py 2>&1 | grep "reduce\|F1 (64,16)\|two-phase (64,16)"; done
