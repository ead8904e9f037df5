# VALU / SALU instructions and wave cycles of the one-LP-per-wavefront Chebyshev kernel (counters-only rocprofv3 pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/wide_pmc; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT -o p -- python $OLDPWD/scripts/debug/wide_check.py > $OUT/log.txt 2>&1)
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/wide_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "cheby_w_kernel" in k or "cheby_r_kernel" in k:
            acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()})
PY
