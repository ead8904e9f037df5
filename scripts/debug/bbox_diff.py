"""GPU: the bounding boxes of listed polytopes of one trial of scripts/soak_wide.py (same RNG stream) side by side with the
oracle's, every side with full precision.   python scripts/debug/bbox_diff.py <seed> <trial> k [k ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
for trial in range(want + 1):
    d = int(rng.choice([4, 5, 5, 6, 6, 7, 8, 8, 9, 10, 12, 13, 14, 16]))
    m = int(rng.integers(d + 1, 65))
    cls = trial % 5
    B = [int(rng.integers(1, 200)), int(rng.integers(1000, 3000)), int(rng.integers(4000, 9000)),
         int(rng.integers(12000, 22000)), int(rng.integers(300, 1000))][cls]
    if d >= 12 or m > 48:
        B = min(B, 6000)
    fam = fams[int(rng.integers(0, len(fams)))]
    A, b, mrows = SL.make(rng, B, m, d, fam)
from oracle import oracle as O
import torch, polytope_amd as pa
nq = min(B, 1500)
bb = pa.bbox_batch(torch.as_tensor(A[:nq]).cuda(), torch.as_tensor(b[:nq]).cuda(), torch.as_tensor(mrows[:nq]).cuda())
lb, ub, st = bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy(), bb["status"].cpu().numpy()
for k in [int(x) for x in sys.argv[3:]]:
    lo, hi, bad = O.bounding_box(A[k, :mrows[k]], b[k, :mrows[k]])
    print("polytope", k, "status", st[k], "oracle bad", bad)
    for i in range(d):
        for nm, mine, ora in (("lb", lb[k][i], lo[i]), ("ub", ub[k][i], hi[i])):
            flag = "" if (mine == ora or abs(mine - ora) <= 1e-9 * 3) else "   <<<<"
            print("  %s[%2d]  kernel %-26r oracle %-26r diff %.3e%s" % (nm, i, float(mine), float(ora), mine - ora if np.isfinite(mine) and np.isfinite(ora) else np.nan, flag))
