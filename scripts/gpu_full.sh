#!/bin/bash
# Full GPU check: the -m gpu suite with per-test durations, smoke, the bench line as the driver runs it.
# Usage: gpurun --timeout 1700 -- 'bash scripts/gpu_full.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
t0=$(date +%s)
timeout 1300 python -m pytest tests -m gpu -x -q --durations=60 2>&1 | tail -90 > gpurun_out/pytest_gpu.log
echo "pytest wall: $(( $(date +%s) - t0 )) s" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"],
      "parity", d.get("parity_checked"), d.get("parity_ok"), "regions", d["config"]["region_ms"])
PY
