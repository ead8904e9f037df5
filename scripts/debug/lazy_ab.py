"""Fused reduce at large shapes: F3 / F2 without a stored dictionary (PLP_REDUCE_LAZY=1, plp_lazy.hpp) against the dense
engines (two rows per lane; PLP_REDUCE_R1=1: one row per lane, the arithmetic twin) -- outputs bitwise, times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, env):
    for k in ("PLP_REDUCE_R1", "PLP_REDUCE_LAZY"): os.environ.pop(k, None)
    os.environ.update(env)
    res = pa.reduce_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.reduce_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) / 5
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(5000, 64, 16), (20000, 64, 16), (5000, 64, 12), (20000, 64, 12), (5000, 48, 9), (20000, 48, 9), (5000, 40, 10), (3000, 64, 13)]
for (B, m, d) in shapes:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    r0, t0 = run(A, b, {})
    r1, t1 = run(A, b, {"PLP_REDUCE_R1": "1"})
    r2, t2 = run(A, b, {"PLP_REDUCE_LAZY": "1"})
    same1 = all(torch.equal(r1[k].view(torch.uint8), r2[k].view(torch.uint8)) for k in r0)
    same0 = all(torch.equal(r0[k].view(torch.uint8), r2[k].view(torch.uint8)) for k in r0)
    nlp = int(r0["nlp"].sum())
    retry = int((r2["flags"] < 0).sum()) if r2["flags"].dtype.is_signed else 0
    print("B=%d m=%d d=%d: dense %.3f ms (%.3g LP/s) | one row/lane %.3f ms | lazy %.3f ms (%.3g LP/s) | lazy == one-row %s, == dense %s"
          % (B, m, d, t0, nlp / t0 * 1e3, t1, t2, nlp / t2 * 1e3, same1, same0), flush=True)
    if not same1:
        for k in r0:
            bad = (r1[k].reshape(B, -1) != r2[k].reshape(B, -1)).any(dim=1).nonzero().flatten()
            print("   ", k, "differs at", bad[:8].tolist(), "count", int(bad.numel()))
