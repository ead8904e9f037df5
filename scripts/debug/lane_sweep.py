"""GPU: reduce_lane_kernel -- share of 8-polytope tiles at the end of a large launch (PLP_REDUCE_LANE_MIX = k / 64), tile
shape against batch size, and the small-batch end against the lane-group latency form."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
dev = torch.device("cuda:0")
NB = 6
full = [random_hpolytopes(100000, 16, 3, seed=i) for i in range(NB)]


def t(B, env, n=60):
    for k in [k for k in os.environ if k.startswith("PLP_REDUCE_")]:
        del os.environ[k]
    os.environ.update(env)
    devb = [(torch.as_tensor(A_[:B]).to(dev), torch.as_tensor(b_[:B]).to(dev)) for A_, b_ in full]
    for k in range(5):
        pa.reduce_batch(*devb[k % NB])
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            pa.reduce_batch(*devb[k % NB])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


for B in (100000, 80000, 60000, 50000, 40000):
    print("B=%6d mix k/64 (us):" % B, " ".join("%d:%.1f" % (k, t(B, {"PLP_REDUCE_LANE_MIX": str(k)})) for k in (0, 2, 4, 6, 8, 12, 16, 24, 32, 48)), flush=True)
for B in (40000, 30000, 20000, 16000, 12000, 8000, 5000, 3000, 2000, 1000, 500, 200, 64, 16, 1):
    print("B=%6d  groups %.1f  GS4 %.1f  GS8 %.1f  GS16 %.1f" % (B, t(B, {"PLP_REDUCE_LANE": "0"}), t(B, {"PLP_REDUCE_LANE": "1", "PLP_REDUCE_LANE_GS": "4"}),
          t(B, {"PLP_REDUCE_LANE": "1", "PLP_REDUCE_LANE_GS": "8"}), t(B, {"PLP_REDUCE_LANE": "1", "PLP_REDUCE_LANE_GS": "16"})), flush=True)
