#!/bin/bash
# PMC passes over bench.py's reduce kernel (counters only; one rocprofv3 run per counter set).
# Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_pmc.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
run_set () {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$name" -o p -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$name.log" 2>&1)
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "plp::" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-24:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("PMC %-24s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
}
run_set insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64
run_set cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run_set misc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES
