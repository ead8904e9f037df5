"""Stand-alone batches on polytopes far from the origin (coordinates ~1e5): what the far-vertex rule of the verifier costs there."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import polytope_amd as pa
import soak_lane as SL
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for (B, m, d) in ((100000, 16, 3), (20000, 64, 8), (5000, 64, 16)):
    A, b, mr = SL.make(rng, B, m, d, "random")
    for off in (0.0, 1e5):
        x0 = off * np.ones(d)
        b2 = b + A @ x0
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b2).to(dev)
        c = np.zeros((B, d)); c[:, 0] = 1.0
        ct = torch.as_tensor(c).to(dev)
        res = {}
        for name, fn in (("cheby", lambda: pa.cheby_ball_batch(At, bt)), ("bbox", lambda: pa.bbox_batch(At, bt)), ("lp", lambda: pa.lpsolve_batch(ct, At, bt))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            res[name] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        print("(%d,%d)x%d offset %g: %s  careful LPs of the last call: %d" % (m, d, B, off, res, pa.verify_careful_lps()), flush=True)
