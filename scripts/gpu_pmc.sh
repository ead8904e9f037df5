#!/bin/bash
# PMC passes (counters only; one rocprofv3 run per counter set) over a command.
# Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_pmc.sh [kernel-substring] [python command...]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
KSUB=${1:-reduce_kernel}; shift || true
if [ $# -eq 0 ]; then set -- python "$PWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline; fi
run_set () {
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$name" -o p -- "$@" > "$OLDPWD/gpurun_out/pmc_$name.log" 2>&1)
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$KSUB" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("PMC %-28s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
}
CTRS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" run_set insts "$@"
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" run_set cycles "$@"
CTRS="GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_SMEM_NORM SQ_ACTIVE_INST_MISC" run_set misc "$@"
