"""Chebyshev LPs: lane-group kernels against the one-LP-per-wavefront engine over mid shapes (where to switch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import polytope_amd as pa
from polytope_amd import synth
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
for B in (2000, 20000):
    for d in (5, 6, 7, 8, 9, 10):
        for m in (16, 24, 32, 33, 40, 48, 64):
            A, b = synth.random_hpolytopes(B, m, d, seed=1)
            At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
            os.environ["PLP_CHEBY_WIDE"] = "0"
            ref = pa.cheby_ball_batch(At, bt); t0 = timeit(lambda: pa.cheby_ball_batch(At, bt))
            os.environ["PLP_CHEBY_WIDE"] = "1"
            got = pa.cheby_ball_batch(At, bt); t1 = timeit(lambda: pa.cheby_ball_batch(At, bt))
            eq = bool(torch.equal(ref["status"], got["status"])) and np.array_equal(ref["r"].cpu().numpy(), got["r"].cpu().numpy(), equal_nan=True)  # (unbounded LPs: NaN radii)
            print("B=%5d d=%2d m=%2d  lane-group %.3f ms  wide %.3f ms  %s %s" % (B, d, m, t0 * 1e3, t1 * 1e3, "WIDE" if t1 < t0 else "    ", "" if eq else "DIFF"), flush=True)
