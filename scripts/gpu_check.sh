#!/bin/bash
# Runs on the GPU box via gpurun: GPU parity tests, smoke, bench, rocprof kernel trace.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [quick]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -m3 -E "gfx|Marketing" 
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
if [ "${1:-}" != "quick" ]; then
  echo "== rocprofv3 kernel trace"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o reduce -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1)
  tail -2 gpurun_out/prof_bench.log
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
fi
if [ "${1:-}" != "quick" ]; then
  echo "== rocprofv3 PMC passes (separate runs, counters only)"
  for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_BUSY_CYCLES; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$ctr" -o reduce -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$ctr.log" 2>&1)
    f=$(find gpurun_out/pmc_$ctr -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $ctr <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[r.get("Kernel_Name", "?")[:60]].append(float(r.get("Counter_Value", 0)))
for k, v in acc.items():
    print(sys.argv[2], k, "dispatches", len(v), "mean", sum(v) / len(v))
PY
  done
fi
