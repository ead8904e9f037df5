#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/gpu_profile_all.sh run from gpurun_out/prof_<tag>/ (scratch) into
profiles/<round>/ (tracked):

  <tag>_kernels.csv         per group and kernel: grid, VGPRs, LDS, scratch, calls, avg / min / max ns
                            (rocprofv3 --kernel-trace of `python scripts/bench_configs.py <group>` / `python bench.py`)
  <tag>_pmc.csv             per group, kernel and counter: dispatches, mean (separate counters-only --pmc passes)
  <tag>_bench_configs.jsonl the JSON lines the traced commands printed, each with `profile` = the rows of
                            <tag>_kernels.csv its kernel time can be checked against
  <tag>_reduce_counters.json  bench kernel: measured SQ counters and what follows from them (no derived "x 4 cycles")

Usage: python scripts/summarize_secondary.py r02 <tag>
"""
import csv
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
summ = json.load(open(os.path.join(src, "summary.json")))

with open(os.path.join(dst, tag + "_kernels.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["group", "kernel", "grid_x", "vgpr", "lds_bytes", "scratch_bytes", "calls", "avg_ns", "min_ns", "max_ns"])
    for name in sorted(summ):
        if "_pmc_" in name:
            continue
        for r in summ[name].get("kernels", []):
            if "plp::" in r["kernel"]:
                w.writerow([name, r["kernel"], r["grid_x"], r["vgpr"], r["lds"], r["scratch"], r["calls"],
                            "%.1f" % r["avg_ns"], r["min_ns"], r["max_ns"]])
with open(os.path.join(dst, tag + "_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["group_pass", "kernel", "grid", "counter", "dispatches", "mean"])
    for name in sorted(summ):
        for r in summ[name].get("counters", []):
            if "plp::" in r["kernel"]:
                w.writerow([name, r["kernel"], r["grid"], r["counter"], r["dispatches"], "%.6g" % r["mean"]])

# bench_configs lines + pointer to the profile rows
with open(os.path.join(dst, tag + "_bench_configs.jsonl"), "w") as out:
    for fn in sorted(os.listdir(src)):
        if not fn.endswith("_bench.jsonl"):
            continue
        group = fn[:-len("_bench.jsonl")]
        for ln in open(os.path.join(src, fn)):
            if not ln.startswith("{"):
                continue
            rec = json.loads(ln)
            rec["profile"] = "profiles/%s/%s_kernels.csv rows group=%s (rocprofv3 --kernel-trace of the command that printed this line)" % (rnd, tag, group)
            out.write(json.dumps(rec) + "\n")


def ctr(passname, kernel_sub, counter, full_grid_only=True):
    rows = [r for r in summ.get(passname, {}).get("counters", []) if kernel_sub in r["kernel"] and r["counter"] == counter]
    if not rows:
        return None
    if full_grid_only:
        g = max(int(r["grid"]) for r in rows)
        rows = [r for r in rows if int(r["grid"]) == g]
    return rows[0]["mean"]


# bench kernel: measured counters (passes bench_pmc_1..7 of gpu_profile_all.sh)
K = sys.argv[3] if len(sys.argv) > 3 else "reduce_lane_mix_kernel<3, 16, 4, 8>"   # the bench kernel: tiles of 16 polytopes, the last eighth of them as tiles of 8
vals = {}
for p in sorted(summ):
    if not p.startswith("bench_pmc_"):
        continue
    for r in summ[p].get("counters", []):
        if K in r["kernel"]:
            g = max(int(q["grid"]) for q in summ[p]["counters"] if K in q["kernel"])
            if int(r["grid"]) == g:
                vals[r["counter"]] = r["mean"]
if vals:
    kt = [r for r in summ.get("bench", {}).get("kernels", []) if K in r["kernel"]]
    gfull = max(int(r["grid_x"]) for r in kt)
    kfull = [r for r in kt if int(r["grid_x"]) == gfull][0]
    out = {"kernel": K, "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-parity --regions 1 --min-region-ms 0 under rocprofv3 --pmc <set> (one pass per set); "
                                   "durations: python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity under --kernel-trace --stats",
           "rocprof_avg_kernel_ns": kfull["avg_ns"], "rocprof_calls": kfull["calls"], "vgpr": kfull["vgpr"],
           "lds_bytes": kfull["lds"], "scratch_bytes": kfull["scratch"], "counters": vals}
    d = {}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        d["hbm_read_bytes"] = vals["FETCH_SIZE"] * 1024 * 2   # gfx950: 64 B counted per 128 B request (MI355X_MICROARCH.md, HBM)
        d["hbm_write_bytes"] = vals["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    if "GRBM_GUI_ACTIVE" in vals:
        d["kernel_cycles_per_xcd"] = vals["GRBM_GUI_ACTIVE"] / 8            # summed over the 8 XCDs
        d["effective_clock_GHz"] = d["kernel_cycles_per_xcd"] / kfull["avg_ns"]
    if "SQ_ACTIVE_INST_VALU" in vals and "SQ_INSTS_VALU" in vals:
        d["quad_cycles_per_valu_inst_measured"] = vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over all SIMDs (1024).  The cycles they are divided by:
        # GRBM_GUI_ACTIVE of the same dispatch -- unless that window is visibly not the kernel's (an effective clock beyond the
        # part's 2.4 GHz: the counter spans more than the dispatch; r06c: 8.5 GHz), then the traced duration at 2.4 GHz.
        simd_cycles = 1024 * d["kernel_cycles_per_xcd"]
        d["busy_frac_basis"] = "GRBM_GUI_ACTIVE"
        if not (1.0 <= d["effective_clock_GHz"] <= 2.45):
            simd_cycles = 1024 * kfull["avg_ns"] * 2.4
            d["busy_frac_basis"] = "traced duration at 2.4 GHz (GRBM_GUI_ACTIVE spans more than the dispatch: %.2f GHz)" % d["effective_clock_GHz"]
            d["effective_clock_GHz"] = None
        d["valu_busy_frac_measured"] = vals["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles
        if "SQ_WAVE_CYCLES" in vals:
            d["mean_resident_waves_per_simd"] = vals["SQ_WAVE_CYCLES"] * 4 / simd_cycles
        if "SQ_WAIT_INST_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
            d["wave_cycles_split"] = {k: vals[k] / vals["SQ_WAVE_CYCLES"] for k in
                                      ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in vals}
    typed = {k: vals[k] for k in vals if k.startswith("SQ_INSTS_VALU_")}
    if typed and "SQ_INSTS_VALU" in vals:
        d["valu_mix"] = {k.replace("SQ_INSTS_VALU_", ""): v / vals["SQ_INSTS_VALU"] for k, v in typed.items() if v}
        d["valu_mix"]["untyped (moves, selects, compares, bit ops, DPP)"] = 1 - sum(typed.values()) / vals["SQ_INSTS_VALU"]
    out["derived"] = d
    json.dump(out, open(os.path.join(dst, tag + "_reduce_counters.json"), "w"), indent=1)
    if "hbm_bytes_per_launch" in d:
        latest = {"kernel": K, "hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "hbm_read_bytes": d["hbm_read_bytes"],
                  "hbm_write_bytes": d["hbm_write_bytes"], "rocprof_avg_kernel_ns": kfull["avg_ns"],
                  "valu_insts_per_launch": vals.get("SQ_INSTS_VALU"),
                  "valu_busy_frac_measured": d.get("valu_busy_frac_measured"),
                  "effective_clock_GHz": d.get("effective_clock_GHz"), "busy_frac_basis": d.get("busy_frac_basis"),
                  "source": "profiles/%s/%s_reduce_counters.json" % (rnd, tag)}
        json.dump(latest, open(os.path.join(root, "profiles", "latest_traffic.json"), "w"), indent=1)
    print(json.dumps(out["derived"], indent=1))
print("wrote", dst)
