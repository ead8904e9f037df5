"""bounding boxes of small batches at d >= 5: the kernels in use (lane groups side by side up to d = 8, one wavefront per polytope
beyond) against four wavefronts per polytope (bbox_wsplit_kernel): ms and equality of lb / ub / status."""
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
for (m, d) in [(64, 8), (32, 6), (16, 5), (64, 12), (32, 10), (64, 16), (24, 14)]:
    for B in [1, 64, 500, 1000, 2000, 4000]:
        A, b = synth.random_hpolytopes(B, m, d, seed=3)
        b = b + np.einsum("bij,bj->bi", A, np.random.default_rng(1).standard_normal((B, d)))
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        out = {}
        for name, env in (("now", {"PLP_BBOX_WSPLIT": "0"}), ("one", {"PLP_BBOX_WIDE": "1", "PLP_BBOX_WSPLIT": "0"}), ("four", {"PLP_BBOX_WIDE": "1", "PLP_BBOX_WSPLIT": "1"})):
            for k in ("PLP_BBOX_WIDE", "PLP_BBOX_WSPLIT"): os.environ.pop(k, None)
            os.environ.update(env)
            res = pa.bbox_batch(At, bt)
            out[name] = ({k: v.cpu().numpy() for k, v in res.items()}, timeit(lambda: pa.bbox_batch(At, bt)))
        same = all(np.array_equal(out["one"][0][k], out["four"][0][k], equal_nan=True) for k in out["one"][0])
        close = all(np.allclose(out["now"][0][k], out["four"][0][k], rtol=0, atol=1e-9, equal_nan=True) for k in ("lb", "ub"))
        print("(%d,%d) B=%-5d in use %.4f ms  one wavefront %.4f  four %.4f (x%.2f of in use)  one==four %s  vs in use %s" % (
            m, d, B, out["now"][1], out["one"][1], out["four"][1], out["now"][1] / out["four"][1], same, close), flush=True)
