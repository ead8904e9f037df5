"""m <= 32, d = 5..16: lane-group kernels (PLP_REDUCE_LAZY=0) against one polytope per workgroup (PLP_REDUCE_LAZY=1): time ratios per batch size."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
Bs = [250, 1000, 2000, 4000, 8000, 20000]
print("shape      " + "  ".join("%7d" % B for B in Bs))
for d in range(5, 17):
    for m in (8, 16, 24, 32):
        row = []
        for B in Bs:
            A, b = synth.random_hpolytopes(B, m, d, seed=2)
            At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
            t = {}
            for v in ("0", "1"):
                os.environ["PLP_REDUCE_LAZY"] = v
                t[v] = timeit(lambda: pa.reduce_batch(At, bt))
            row.append(t["0"] / t["1"])
        print("(%2d,%2d)   " % (m, d) + "  ".join("%7.2f" % x for x in row), flush=True)
