# bench line for several shares of half-size tiles at the end of the launch (PLP_REDUCE_MIX = k/64 of the tiles)
cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 4 6 8 12 16 24 32; do
  PLP_REDUCE_MIX=$v python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MIX=$v', '%.4g LP/s' % d['value'], '%.4f ms' % d['ms_per_step'])"
done
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', '%.4g LP/s' % d['value'], '%.4f ms' % d['ms_per_step'])"
