"""Fused reduce at large shapes: one row per lane / one polytope per wavefront (PLP_REDUCE_R1=1) against two rows per
lane, outputs compared bitwise, times per batch size."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, r1):
    if r1: os.environ["PLP_REDUCE_R1"] = "1"
    else: os.environ.pop("PLP_REDUCE_R1", None)
    res = pa.reduce_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): pa.reduce_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) / 10
for (B, m, d) in [(5000, 64, 16), (20000, 64, 16), (5000, 64, 12), (20000, 64, 12), (5000, 48, 9), (20000, 48, 9)]:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    r0, t0 = run(A, b, False)
    r1, t1 = run(A, b, True)
    same = all(torch.equal(r0[k].view(torch.uint8), r1[k].view(torch.uint8)) for k in r0)
    nlp = int(r0["nlp"].sum())
    print("B=%d m=%d d=%d: two rows/lane %.3f ms (%.3g LP/s)  one row/lane %.3f ms (%.3g LP/s)  bitwise equal %s" % (B, m, d, t0, nlp / t0 * 1e3, t1, nlp / t1 * 1e3, same))
