"""Small-batch fused reduce: latency form (one polytope per wavefront, LPs over the lane groups; default for B <= 8192)
against the batch form (PLP_REDUCE_SPLIT=0): outputs bitwise, device time per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, m, split):
    os.environ["PLP_REDUCE_SPLIT"] = "1" if split else "0"
    res = pa.reduce_batch(A, b, m=m); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(10): pa.reduce_batch(A, b, m=m)
    ev[0].record()
    for _ in range(100): pa.reduce_batch(A, b, m=m)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) * 10
rng = np.random.default_rng(3)
for (m, d) in [(16, 3), (12, 4), (16, 2), (10, 1), (32, 6), (24, 5), (64, 8), (40, 7), (16, 8)]:
    for B in (1, 16, 256, 4096, 16384):
        A, b = random_hpolytopes(B, m, d, seed=2 + B, stream=0)
        A[::5, 1] = A[::5, 0]; b[::5, 1] = b[::5, 0] + 0.01     # duplicates (dedupe), some infeasible
        b[3::17, 2] = -3.0
        rows = rng.integers(max(d + 1, m - 4), m + 1, B).astype(np.int32)
        At = torch.as_tensor(A).cuda(); bt = torch.as_tensor(b).cuda(); mt = torch.as_tensor(rows).cuda()
        r0, t0 = run(At, bt, mt, False)
        r1, t1 = run(At, bt, mt, True)
        same = all(torch.equal(r0[k].view(torch.uint8), r1[k].view(torch.uint8)) for k in r0)
        print("(%d,%d) B=%-6d batch form %.1f us, latency form %.1f us, bitwise equal %s" % (m, d, B, t0, t1, same), flush=True)
        if not same:
            for k in r0:
                bad = (r0[k].reshape(B, -1) != r1[k].reshape(B, -1)).any(dim=1).nonzero().flatten()
                if bad.numel(): print("    ", k, "differs at", bad[:6].tolist(), "count", int(bad.numel()))
