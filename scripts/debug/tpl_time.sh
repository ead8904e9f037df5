cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "PLP_TPL=$v"; PLP_TPL=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
