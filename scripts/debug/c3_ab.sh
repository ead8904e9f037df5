cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "contains" 2>&1 | tail -3
PLP_CONTAINS_THR=0 python scripts/bench_configs.py c3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('THR=0', d['ms'], d['roofline']['frac'], d['spot_parity_equal'])"
python scripts/bench_configs.py c3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('THR=1', d['ms'], d['roofline']['frac'], d['spot_parity_equal'])"
