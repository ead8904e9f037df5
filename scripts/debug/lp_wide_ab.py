"""Generic lpsolve batches: lane-group kernels (PLP_LP_WIDE=0) against one LP per wavefront (PLP_LP_WIDE=1), LPs whose
origin is infeasible (two phases) and feasible (one) -- status, iterations, x, times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
rng = np.random.default_rng(0)
def run(ct, At, bt, env):
    if env is None: os.environ.pop("PLP_LP_WIDE", None)
    else: os.environ["PLP_LP_WIDE"] = env
    r = pa.lpsolve_batch(ct, At, bt); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.lpsolve_batch(ct, At, bt)
    ev[1].record(); torch.cuda.synchronize()
    return r, ev[0].elapsed_time(ev[1]) / 5
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [
    (20000, 32, 6), (20000, 24, 5), (20000, 16, 5), (20000, 32, 8), (20000, 20, 10), (20000, 32, 12), (4096, 32, 6), (1000, 32, 6), (1, 32, 6),
    (1, 64, 16), (20000, 64, 8), (20000, 48, 5), (20000, 64, 16)]
for (B, m, n) in shapes:
    for kind in ("two-phase", "feasible"):
        A, b = random_hpolytopes(B, m, n, seed=2, stream=0)
        if kind == "two-phase":
            b = b + np.einsum("bij,bj->bi", A, rng.standard_normal((B, n)) * 3.0)
        c = rng.standard_normal((B, n))
        ct, At, bt = (torch.as_tensor(v).cuda() for v in (c, A, b))
        r0, t0 = run(ct, At, bt, "0")
        r1, t1 = run(ct, At, bt, "1")
        rd, td = run(ct, At, bt, None)
        assert torch.equal(rd["status"], r1["status"]) and np.array_equal(rd["x"].cpu().numpy(), r1["x"].cpu().numpy(), equal_nan=True)
        st = torch.equal(r0["status"], r1["status"])
        it = torch.equal(r0["iters"], r1["iters"]) if "iters" in r0 else None
        x0, x1 = r0["x"].cpu().numpy(), r1["x"].cpu().numpy()
        bit = float((x0.view(np.uint64) == x1.view(np.uint64)).all(axis=1).mean())
        close = np.allclose(x0, x1, rtol=1e-9, atol=1e-9, equal_nan=True)
        print("B=%5d m=%2d n=%2d %-9s: lane groups %.3f ms | wide %.3f ms (%.3g LP/s) %s | default %.3f | status equal %s iters equal %s x close %s (bitwise %.3f), statuses %s"
              % (B, m, n, kind, t0, t1, B / t1 * 1e3, "WIDE" if t1 < t0 else "    ", td, st, it, close, bit, np.unique(r1["status"].cpu().numpy()).tolist()), flush=True)
os.environ.pop("PLP_LP_WIDE", None)
