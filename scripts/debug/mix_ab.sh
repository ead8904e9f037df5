cd $GRAFT_REPO_ROOT
for v in 0 4; do
  PLP_REDUCE_MIX=$v python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MIX=$v', '%.4g LP/s' % d['value'], '%.4f ms' % d['ms_per_step'])"
done
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', '%.4g LP/s' % d['value'], '%.4f ms' % d['ms_per_step'])"
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-end-to-end --pipelined 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['pipelined'])"
timeout 1400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
