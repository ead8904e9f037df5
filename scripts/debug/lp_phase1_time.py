"""Generic lpsolve batches whose origin is infeasible (two phases), by shape: LP/s per engine."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
rng = np.random.default_rng(0)
for (B, m, n) in [(100000, 16, 3), (20000, 32, 6), (20000, 64, 8), (20000, 64, 12), (20000, 64, 16), (2000, 64, 16), (20000, 40, 10)]:
    A, b = random_hpolytopes(B, m, n, seed=2, stream=0)
    cen = rng.standard_normal((B, n)) * 3.0
    b = b + np.einsum("bij,bj->bi", A, cen)            # the polytope moved away from the origin: phase 1 needed
    c = rng.standard_normal((B, n))
    ct, At, bt = (torch.as_tensor(v).cuda() for v in (c, A, b))
    r = pa.lpsolve_batch(ct, At, bt); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.lpsolve_batch(ct, At, bt)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    st = r["status"].cpu().numpy()
    print("B=%6d m=%2d n=%2d: %.3f ms  %.3g LP/s  (status 0: %d, other: %d; mean iterations %.1f)" % (B, m, n, ms, B / ms * 1e3, int((st == 0).sum()), int((st != 0).sum()), float(r["iters"].float().mean()) if "iters" in r else -1), flush=True)
