#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/ (scratch) into profiles/<round>/ (tracked):
kernel stats of the rocprofv3 --kernel-trace --stats run, the two PMC passes, the bench line,
and traffic.json = HBM bytes per launch of the dominant kernel
(FETCH_SIZE [KB] x 1024 x 2 -- gfx950 counts 64 B per 128 B read request, MI355X_MICROARCH.md
section HBM -- plus WRITE_SIZE [KB] x 1024)."""
import csv
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "prof", "reduce_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.log"), os.path.join(dst, tag + "_bench.json"))
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"):
    path = os.path.join(src, "pmc_" + ctr, "reduce_counter_collection.csv")
    if not os.path.exists(path):
        continue
    rows = list(csv.DictReader(open(path)))
    mine = [r for r in rows if "plp::" in r["Kernel_Name"]]
    with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, ctr)), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatch_Id", "Counter_Name", "Counter_Value", "VGPR_Count", "LDS_Block_Size", "Scratch_Size"])
        for r in mine:
            w.writerow([r["Kernel_Name"].split("(")[0], r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"],
                        r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")])
    # full-size dispatches only (bench.py primes the code object with a 64-polytope call)
    rk = [r for r in mine if "plp::reduce_r_kernel" in r["Kernel_Name"]]
    gmax = max(int(r["Grid_Size"]) for r in rk)
    v = [float(r["Counter_Value"]) for r in rk if int(r["Grid_Size"]) == gmax]
    vals[ctr] = sum(v) / len(v)
stats = list(csv.DictReader(open(os.path.join(src, "prof", "reduce_kernel_stats.csv"))))
k = dict([r for r in stats if "plp::reduce_r_kernel" in r["Name"]][0])
k_name = k["Name"].split("(")[0].replace("void ", "")
# the stats line averages over every dispatch, including the 64-polytope priming call: redo the average over the
# full-size dispatches from the kernel trace of the same run
ktr = [r for r in csv.DictReader(open(os.path.join(src, "prof", "reduce_kernel_trace.csv")))
       if "plp::reduce_r_kernel" in r["Kernel_Name"]]
gfull = max(int(r["Grid_Size_X"]) for r in ktr)
dfull = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ktr if int(r["Grid_Size_X"]) == gfull]
k["AverageNs"], k["Calls"] = sum(dfull) / len(dfull), len(dfull)
shutil.copy(os.path.join(src, "prof", "reduce_kernel_trace.csv"), os.path.join(dst, tag + "_kernel_trace.csv"))
traffic = {
    "kernel": k_name,
    "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (one rocprofv3 --pmc run per counter)",
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "hbm_read_bytes": vals["FETCH_SIZE"] * 1024 * 2, "hbm_write_bytes": vals["WRITE_SIZE"] * 1024,
    "hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024,
    "rocprof_avg_kernel_ns": float(k["AverageNs"]), "rocprof_calls": int(k["Calls"]),
}
if "SQ_INSTS_VALU" in vals:
    # every VALU instruction of a wave64 occupies its SIMD's issue port for 4 cycles; 256 CUs x 4 SIMDs
    traffic["valu_insts_per_launch"] = vals["SQ_INSTS_VALU"]
    traffic["valu_issue_cycles_per_simd"] = vals["SQ_INSTS_VALU"] * 4 / 1024
    traffic["valu_issue_frac_at_2p4GHz"] = vals["SQ_INSTS_VALU"] * 4 / 1024 / (float(k["AverageNs"]) * 2.4)
if "SQ_BUSY_CYCLES" in vals and "SQ_INSTS_VALU" in vals:
    # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4); per engine it is the kernel's duration in
    # shader clocks, which gives the clock the chip held (kernel-trace timestamps of the same PMC pass) and the
    # VALU issue fraction without assuming 2.4 GHz
    tr = list(csv.DictReader(open(os.path.join(src, "pmc_SQ_BUSY_CYCLES", "reduce_kernel_trace.csv"))))
    gm = max(int(r["Grid_Size_X"]) for r in tr if "plp::reduce_r_kernel" in r["Kernel_Name"])
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr
           if "plp::reduce_r_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == gm]
    per_se = vals["SQ_BUSY_CYCLES"] / 32
    traffic["sq_busy_cycles_per_shader_engine"] = per_se
    traffic["shader_clock_GHz_estimate"] = per_se / (sum(dur) / len(dur))
    traffic["valu_issue_frac_of_busy_cycles"] = vals["SQ_INSTS_VALU"] * 4 / 1024 / per_se
json.dump(traffic, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(root, "profiles", "latest_traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
