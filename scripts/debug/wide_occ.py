"""F1 on the one-LP-per-wavefront engine against the number of resident wavefronts (PLP_WIDE_LDSPAD bytes of unused LDS per
workgroup): run once per pad value.   PLP_WIDE_LDSPAD=7800 python scripts/debug/wide_occ.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth
dev = torch.device("cuda:0")
def timeit(fn, reps=9, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = []
for (B, m, d) in [(20000, 64, 8), (80000, 64, 8), (20000, 64, 12), (20000, 64, 16), (80000, 64, 16)]:
    A, b = synth.random_hpolytopes(B, m, d, seed=1)
    At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
    out.append("(%d,%d)x%d %.4f ms" % (m, d, B, timeit(lambda: pa.cheby_ball_batch(At, bt))))
print("pad", os.environ.get("PLP_WIDE_LDSPAD", "0"), " | ".join(out))
