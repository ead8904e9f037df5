"""Fused reduce, one polytope per wavefront: F3 / F2 on the dense one-LP-per-wavefront engine (PLP_REDUCE_WDENSE=1) against
the same kernel without a stored dictionary (=0) and against the default dispatch -- outputs bitwise, times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, env):
    for k in ("PLP_REDUCE_R1", "PLP_REDUCE_LAZY", "PLP_REDUCE_WDENSE"): os.environ.pop(k, None)
    os.environ.update(env)
    res = pa.reduce_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.reduce_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) / 5
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [
    (5000, 64, 8), (20000, 64, 8), (20000, 48, 6), (20000, 33, 5), (20000, 64, 5), (5000, 40, 7), (20000, 32, 6), (20000, 32, 8),
    (5000, 64, 16), (20000, 64, 16), (5000, 64, 12), (20000, 64, 12), (20000, 48, 9), (5000, 40, 10), (20000, 32, 12)]
for (B, m, d) in shapes:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    r0, t0 = run(A, b, {})
    r2, t2 = run(A, b, {"PLP_REDUCE_LAZY": "1", "PLP_REDUCE_WDENSE": "0"})
    r3, t3 = run(A, b, {"PLP_REDUCE_LAZY": "1", "PLP_REDUCE_WDENSE": "1"})
    same = all(torch.equal(r3[k].view(torch.uint8), r2[k].view(torch.uint8)) for k in r0)
    same0 = all(torch.equal(r3[k].view(torch.uint8), r0[k].view(torch.uint8)) for k in ("keep", "flags", "nlp", "r"))
    nlp = int(r0["nlp"].sum())
    best = min((t0, "default"), (t2, "lazy"), (t3, "wdense"))[1]
    print("B=%5d m=%2d d=%2d: default %.3f ms (%.3g LP/s) | lazy %.3f | wdense %.3f (%.3g LP/s) | best %-7s | wdense == lazy %s, == default (keep, flags, nlp, r) %s"
          % (B, m, d, t0, nlp / t0 * 1e3, t2, t3, nlp / t3 * 1e3, best, same, same0), flush=True)
    if not same:
        for k in r0:
            bad = (r3[k].reshape(B, -1) != r2[k].reshape(B, -1)).any(dim=1).nonzero().flatten()
            if bad.numel(): print("   ", k, "differs at", bad[:8].tolist(), "count", int(bad.numel()))
