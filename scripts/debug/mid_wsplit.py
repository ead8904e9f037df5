"""(m <= 32, d = 5..8): the lane-group kernels (default) against one polytope per workgroup (PLP_REDUCE_LAZY=1 -> reduce_wsplit_kernel)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth
def timeit(fn, reps=9):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for (m, d) in [(32, 12), (30, 9), (32, 16), (24, 14), (12, 5), (16, 8)]:
    for B in [1, 64, 250, 1000, 2000, 4000, 8000, 20000]:
        A, b = synth.random_hpolytopes(B, m, d, seed=2)
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        out = {}
        for v in ("", "1"):
            if v: os.environ["PLP_REDUCE_LAZY"] = v
            else: os.environ.pop("PLP_REDUCE_LAZY", None)
            res = pa.reduce_batch(At, bt)
            out[v] = ({k: x.cpu().numpy() for k, x in res.items()}, timeit(lambda: pa.reduce_batch(At, bt)))
        os.environ.pop("PLP_REDUCE_LAZY", None)
        same = all(np.array_equal(out[""][0][k], out["1"][0][k]) for k in ("keep", "flags", "nlp"))
        print("(%d,%d) B=%-5d lane groups %.4f ms  per workgroup %.4f ms (x%.2f)  masks %s" % (m, d, B, out[""][1], out["1"][1], out[""][1] / out["1"][1], "equal" if same else "DIFFERENT"), flush=True)
