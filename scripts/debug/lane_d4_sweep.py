"""Fused reduce at d = 4: the lane kernel (walk4; PLP_REDUCE_LANE=1: at every batch size) against the lane-group kernels
(PLP_REDUCE_LANE=0), us per launch -- for the dispatch threshold.   python scripts/debug/lane_d4_sweep.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for m in (8, 12, 16, 20, 24, 32):
    row = []
    for B in (500, 2000, 5000, 10000, 20000, 30000, 50000):
        A, b = random_hpolytopes(B, m, 4, seed=m)
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        out = []
        for lane in ("0", "1"):
            os.environ["PLP_REDUCE_LANE"] = lane
            out.append(t(lambda: pa.reduce_batch(At, bt)))
        row.append("%d: %.0f/%.0f" % (B, out[0], out[1]))
    print("(%d,4) groups/lane us  " % m + "  ".join(row), flush=True)
