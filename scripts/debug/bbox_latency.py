import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (m, d) in [(16, 3), (12, 4), (32, 6), (64, 8)]:
    for B in (1, 64, 1024):
        A, b = random_hpolytopes(B, m, d, seed=5, stream=0, bounded=True)
        At = torch.as_tensor(A).cuda(); bt = torch.as_tensor(b).cuda()
        for _ in range(10): pa.bbox_batch(At, bt)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(100): pa.bbox_batch(At, bt)
        ev[1].record(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): pa.bbox_batch(A, b)
        t1 = time.perf_counter()
        print("(%d,%d) B=%-5d bbox device %.1f us, numpy call %.1f us" % (m, d, B, ev[0].elapsed_time(ev[1]) * 10, (t1 - t0) * 1e4))
