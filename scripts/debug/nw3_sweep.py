import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import polytope_amd as pa
from polytope_amd import synth
def timeit(fn, reps=9):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for (m, d) in [(64, 12), (48, 10), (64, 8)]:
    for B in [2000, 3000, 5000, 8000, 12000, 16000]:
        A, b = synth.random_hpolytopes(B, m, d, seed=2)
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        row = []
        for v in ("2", "4", "4"):
            os.environ["PLP_REDUCE_WSPLIT"] = v
            row.append(timeit(lambda: pa.reduce_batch(At, bt)))
        print("(%d,%d) B=%d  NW=2 %.4f  NW=3 %.4f  NW=4 %.4f" % (m, d, B, *row), flush=True)
