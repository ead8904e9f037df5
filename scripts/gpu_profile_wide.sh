#!/bin/bash
# SQ counter sets (instructions, busy / wait split, resident waves) for the kernels of the LARGER shapes: the `lp` and `red`
# groups of scripts/bench_configs.py (cheby_w_kernel, lp_w_kernel, reduce_wdense_kernel, reduce_r_kernel<D,16,2>, ...), one
# counters-only rocprofv3 pass per set, plus one --kernel-trace pass for the durations.  Runs on the GPU box:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_profile_wide.sh <tag> [groups...]'
# writes gpurun_out/wide_<tag>/wide_counters.json (per kernel: mean of every counter over its dispatches + derived fractions);
# copy it to profiles/rNN/<tag>_wide_counters.json.
set -u
TAG=${1:-a}; shift || true
GROUPS_=${*:-"lp red"}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/wide_$TAG
mkdir -p "$OUT"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python "$OLDPWD/scripts/bench_configs.py" $GROUPS_ > "$OUT/bench.jsonl" 2> "$OUT/kt.err")
i=0
for set_ in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
    "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
    "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- \
      python "$OLDPWD/scripts/bench_configs.py" $GROUPS_ > /dev/null 2> "$OUT/pmc_$i.err")
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
dur = collections.defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "plp::" not in k:
            continue
        key = (k, r["Grid_Size_X"])
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[key] = dict(vgpr=r.get("VGPR_Count"), accum_vgpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                         lds=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"), wg=r.get("Workgroup_Size_X"))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "plp::" not in k:
            continue
        ctr[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = []
for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    c = {n: sum(x) / len(x) for n, x in ctr.get(key, {}).items()}
    d = dict(kernel=key[0], grid_x=int(key[1]), calls=len(v), avg_ns=sum(v) / len(v), **meta[key], counters=c)
    if "SQ_ACTIVE_INST_VALU" in c:
        for name, simd_cycles in (("by_duration_at_2.4GHz", 1024 * d["avg_ns"] * 2.4),
                                  ("by_grbm", 1024 * c.get("GRBM_GUI_ACTIVE", 0) / 8)):
            if simd_cycles <= 0:
                continue
            e = {"valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles}
            if "SQ_WAVE_CYCLES" in c:
                e["mean_resident_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / simd_cycles
            if "SQ_ACTIVE_INST_SCA" in c:
                e["salu_busy_frac"] = c["SQ_ACTIVE_INST_SCA"] * 4 / simd_cycles
            if "SQ_ACTIVE_INST_LDS" in c:
                e["lds_busy_frac"] = c["SQ_ACTIVE_INST_LDS"] * 4 / simd_cycles
            d[name] = e
        if "SQ_WAVE_CYCLES" in c:
            d["wave_cycles_split"] = {n: c[n] / c["SQ_WAVE_CYCLES"] for n in
                                      ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS") if n in c}
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        d["per_wave"] = {n.replace("SQ_INSTS_", "").lower(): c[n] / c["SQ_WAVES"] for n in c if n.startswith("SQ_INSTS_")}
    res.append(d)
json.dump(res, open(os.path.join(out, "wide_counters.json"), "w"), indent=1)
for d in res[:14]:
    print("%-46s grid %-8d %8.1f us  vgpr %s busy %s waves %s  per-wave valu/salu/lds %s" % (
        d["kernel"][-46:], d["grid_x"], d["avg_ns"] / 1e3, d["vgpr"],
        "%.2f" % d.get("by_duration_at_2.4GHz", {}).get("valu_busy_frac", float("nan")),
        "%.2f" % d.get("by_duration_at_2.4GHz", {}).get("mean_resident_waves_per_simd", float("nan")),
        [round(d.get("per_wave", {}).get(k, 0)) for k in ("valu", "salu", "lds")]))
PY
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
