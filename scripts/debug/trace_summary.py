"""Per-kernel summary of a rocprofv3 --kernel-trace csv directory: python scripts/debug/trace_summary.py <dir> [substring ...]"""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("plp::", "")
        acc[(k[:64], r["Grid_Size_X"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
want = sys.argv[2:]
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if not want or any(w in k[0] for w in want):
        print("%-66s grid %-9s vgpr %-4s lds %-6s scratch %-5s calls %3d avg us %9.1f" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v) / 1e3))
