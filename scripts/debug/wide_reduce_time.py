"""Fused reduce beyond 64 rows (reduce_lds_kernel): time per batch by shape, beside the 64-row kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (B, m, d) in [(100, 69, 4), (1000, 69, 4), (1000, 80, 3), (1000, 100, 4), (1000, 128, 8), (1000, 64, 4), (1000, 64, 8), (100, 64, 4), (1, 69, 4), (1, 64, 4), (5000, 100, 6)]:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    res = pa.reduce_batch(At, bt); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.reduce_batch(At, bt)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    nlp = int(res["nlp"].sum())
    print("B=%5d m=%3d d=%d: %.3f ms  %.3g LP/s (%d LPs, %.1f per polytope)" % (B, m, d, ms, nlp / ms * 1e3, nlp, nlp / B), flush=True)
