cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "chunked_upload or reject_inf or lpsolve or cheby or bbox" 2>&1 | tail -3
for t in 0 1; do
PLP_STAGE=$t python - <<'PY'
import os, time, numpy as np, sys
sys.path.insert(0, '.')
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
A, b = random_hpolytopes(100000, 16, 3, seed=0, stream=0)
res = pa.reduce_batch(A, b); nlp = int(res["nlp"].sum())
for _ in range(3): pa.reduce_batch(A, b)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); pa.reduce_batch(A, b); ts.append(time.perf_counter() - t0)
ts.sort()
print("PLP_STAGE=%s median %.3f ms min %.3f ms -> %.3g LP/s" % (os.environ.get("PLP_STAGE"), ts[7] * 1e3, ts[0] * 1e3, nlp / ts[7]))
PY
done
