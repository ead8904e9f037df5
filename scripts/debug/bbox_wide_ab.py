"""Fused bounding boxes: lane-group kernels (PLP_BBOX_WIDE=0) against one polytope per wavefront with wave-uniform pivots
(PLP_BBOX_WIDE=1; dense / lazy 2d LPs by PLP_BBOX_WDENSE) -- values, status, times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, env):
    for k in ("PLP_BBOX_WIDE", "PLP_BBOX_WDENSE"): os.environ.pop(k, None)
    os.environ.update(env)
    res = pa.bbox_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.bbox_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) / 5
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [
    (5000, 64, 8), (20000, 64, 8), (20000, 48, 6), (20000, 33, 5), (20000, 32, 6), (2000, 32, 6), (1000, 64, 8), (100, 64, 8), (1, 64, 8),
    (1, 32, 6), (20000, 24, 5), (5000, 64, 12), (20000, 48, 9), (5000, 64, 16), (5000, 64, 14)]
for (B, m, d) in shapes:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    r0, t0 = run(A, b, {"PLP_BBOX_WIDE": "0"} if d <= 8 else {"PLP_BBOX_WDENSE": "0"})
    r1, t1 = run(A, b, {"PLP_BBOX_WIDE": "1", "PLP_BBOX_WDENSE": "1"})
    rd, td = run(A, b, {})
    ok = all(np.allclose(r0[k].cpu().numpy(), r1[k].cpu().numpy(), rtol=0, atol=1e-10, equal_nan=True) for k in ("lb", "ub"))
    st = torch.equal(r0["status"], r1["status"])
    print("B=%5d m=%2d d=%2d: lane groups / lazy %.3f ms | wide dense %.3f ms (%.3g LP/s) | default %.3f | values equal %s status equal %s (handed back %d / %d)"
          % (B, m, d, t0, t1, B * (2 * d + 1) / t1 * 1e3, td, ok, st, int((r0["status"] != 0).sum()), int((r1["status"] != 0).sum())), flush=True)
