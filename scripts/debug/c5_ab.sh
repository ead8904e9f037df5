cd $GRAFT_REPO_ROOT
for cap in default 512 2048 4096 8192; do
  if [ $cap = default ]; then unset PLP_ASSIGN_BLOCKS; else export PLP_ASSIGN_BLOCKS=$cap; fi
  echo "cap=$cap"; timeout 200 python scripts/bench_configs.py c5 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['config'],'ms %.4f'%d['ms'],d['facet_ids_equal_numpy'])"
done
