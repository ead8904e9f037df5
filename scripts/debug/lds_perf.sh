cd $GRAFT_REPO_ROOT
echo default; python scripts/bench_configs.py lp 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], '%.3f ms' % d['ms'], '%.3g LP/s' % d['lp_per_s'])"
echo PLP_LDS=1; PLP_LDS=1 python scripts/bench_configs.py lp 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], '%.3f ms' % d['ms'], '%.3g LP/s' % d['lp_per_s'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_force" 2>&1 | tail -3
