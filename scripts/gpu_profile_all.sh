#!/bin/bash
# rocprofv3 evidence for EVERY kernel of the path, not only the bench kernel (runs on the GPU box):
#   per config group of scripts/bench_configs.py: one --kernel-trace --stats run (per-kernel durations) and
#   separate --pmc passes (FETCH_SIZE, WRITE_SIZE: HBM traffic per launch), counters only;
#   for bench.py (the bench kernel): measured SQ cycle / instruction counters instead of derived ones.
# Usage: gpurun --timeout 2400 -- 'bash scripts/gpu_profile_all.sh <tag> [groups...]'
#        then  python scripts/summarize_secondary.py r02 <tag>   copies the summaries into profiles/r02/.
set -u
TAG=${1:-a}; shift || true
GROUPS_=${*:-"c3 c5 lp red bbox c4 hull"}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
(cd /tmp && rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1)
for g in $GROUPS_; do
  echo "== group $g: kernel trace"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$g" -o kt -- \
      python "$OLDPWD/scripts/bench_configs.py" $g > "$OUT/${g}_bench.jsonl" 2> "$OUT/${g}_trace.err")
  grep -h '^{' "$OUT/${g}_bench.jsonl" | cut -c1-300
  for ctr in FETCH_SIZE WRITE_SIZE; do
    case $g in hull|c4|bbox) continue;; esac
    (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/${g}_pmc_$ctr" -o p -- \
        python "$OLDPWD/scripts/bench_configs.py" $g > /dev/null 2> "$OUT/${g}_pmc_$ctr.err")
  done
done
if [[ " $GROUPS_ " == *" bench "* ]]; then
  echo "== bench.py: kernel trace + SQ counters (one pass per set)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o kt -- \
      python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-secondary > "$OUT/bench_trace.log" 2>&1)
  tail -1 "$OUT/bench_trace.log" | cut -c1-400
  i=0
  for set_ in "FETCH_SIZE" "WRITE_SIZE" \
      "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
      "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
      "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" \
      "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    (cd /tmp && timeout 900 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d "$OUT/bench_pmc_$i" -o p -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-parity --no-secondary --regions 1 --min-region-ms 0 > "$OUT/bench_pmc_$i.log" 2>&1)
  done
fi
# compact summaries (the raw csv files of a trace run are large)
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
summ = {}
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(k, r["Grid_Size_X"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        rows = [dict(kernel=k[0], grid_x=k[1], vgpr=k[2], lds=k[3], scratch=k[4], calls=len(v), avg_ns=sum(v) / len(v), min_ns=min(v), max_ns=max(v)) for k, v in acc.items()]
        summ.setdefault(name, {})["kernels"] = sorted(rows, key=lambda r: -r["avg_ns"] * r["calls"])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(k, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        rows = [dict(kernel=k[0], grid=k[1], counter=k[2], dispatches=len(v), mean=sum(v) / len(v)) for k, v in acc.items()]
        summ.setdefault(name, {})["counters"] = rows
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
for name, s in summ.items():
    for r in s.get("kernels", [])[:6]:
        if "plp::" in r["kernel"]:
            print("KT %-14s %-44s grid %-9s calls %-4d avg %.1f us" % (name, r["kernel"][-44:], r["grid_x"], r["calls"], r["avg_ns"] / 1e3))
PY
# keep what travels back small: drop the raw per-dispatch csv files except the stats
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
