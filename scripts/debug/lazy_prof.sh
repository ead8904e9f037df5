cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lazy_prof; rm -rf $OUT; mkdir -p $OUT
i=0
for set_ in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $OLDPWD/scripts/debug/lazy_prof.py > $OUT/p$i.log 2>&1)
done
python - $OUT <<'PY'
import csv, glob, sys, collections, os
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reduce" in k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-44s %-22s %.4g" % (k[-44:], c, sum(v) / len(v)))
PY
