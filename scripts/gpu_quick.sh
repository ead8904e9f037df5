#!/bin/bash
# Quick GPU iteration: the reduce parity tests + the bench line (no CPU baseline leg).
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_quick.sh [pytest -k expression]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
K="${1:-reduce or smoke or fuzz}"
timeout 800 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -15 | tee gpurun_out/quick_pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step',d['ms_per_step'],'kernel_ms',r['kernel_ms'],'value',d['value'])" | tee -a gpurun_out/quick_bench.log; done
