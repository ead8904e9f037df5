"""GPU: the tail of a large reduce_lane launch -- share of 8-polytope tiles (MIX = k / 64 of the 16-polytope tiles) and, of
those, the share turned into 4-polytope tiles (MIX3 = k / 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
dev = torch.device("cuda:0")
NB = 6
full = [random_hpolytopes(100000, 16, 3, seed=i) for i in range(NB)]


def t(B, env, n=80):
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


for B in (100000, 75000, 50000):
    for mix in (4, 8, 12, 16):
        print("B=%6d MIX %2d  MIX3 (us):" % (B, mix), " ".join("%d:%.1f" % (k, t(B, {"PLP_REDUCE_LANE_MIX": str(mix), "PLP_REDUCE_LANE_MIX3": str(k)})) for k in (0, 8, 16, 32, 48, 64)), flush=True)
