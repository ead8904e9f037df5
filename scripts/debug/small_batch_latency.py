"""Latency of small fused-reduce / Chebyshev batches (device tensors in, events around 200 calls)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (m, d) in [(16, 3), (12, 4), (32, 6)]:
    for B in (1, 16, 256, 4096):
        A, b = random_hpolytopes(B, m, d, seed=2, stream=0)
        At = torch.as_tensor(A).cuda(); bt = torch.as_tensor(b).cuda()
        for name, fn in (("reduce", lambda: pa.reduce_batch(At, bt)), ("cheby", lambda: pa.cheby_ball_batch(At, bt))):
            for _ in range(20): fn()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            t0 = time.perf_counter(); ev[0].record()
            for _ in range(200): fn()
            ev[1].record(); torch.cuda.synchronize(); t1 = time.perf_counter()
            # one call at a time (launch + wait): what a Python loop over single polytopes sees
            t2 = time.perf_counter()
            for _ in range(200): fn(); torch.cuda.synchronize()
            t3 = time.perf_counter()
            print("(%d,%d) B=%-5d %-6s back-to-back %.1f us/call (device %.1f), call+sync %.1f us" % (m, d, B, name, (t1 - t0) / 200 * 1e6, ev[0].elapsed_time(ev[1]) / 200 * 1e3, (t3 - t2) / 200 * 1e6))
