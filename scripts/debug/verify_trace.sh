#!/bin/bash
# The verifier's smoke (parity against the oracle + timings) and a kernel trace of it.
# Usage: gpurun --timeout 900 -- 'bash scripts/debug/verify_trace.sh'
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python scripts/debug/verify_smoke.py 0 time > gpurun_out/vs.log 2>&1
grep -E "SMOKE|TIMING|mismatch" gpurun_out/vs.log | tail -4
rm -rf /tmp/tr
timeout 400 rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python scripts/debug/verify_smoke.py 0 time > /dev/null 2>&1
python scripts/debug/trace_summary.py /tmp/tr verify careful > gpurun_out/tr.log 2>&1
cat gpurun_out/tr.log
