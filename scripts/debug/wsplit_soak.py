"""Repeat the small-batch reduce launches (reduce_wsplit_kernel: wavefronts of a workgroup meet in LDS atomics and barriers) and
compare every output with the first run's: any difference is a race.   python scripts/debug/wsplit_soak.py [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (B, m, d) in [(300, 64, 8), (3000, 64, 8), (1400, 48, 6), (250, 64, 16), (2500, 64, 14), (900, 40, 11), (12000, 33, 5)]:
    A, b = synth.random_hpolytopes(B, m, d, seed=9)
    rng = np.random.default_rng(1)
    for k in range(0, B, 5):
        j = rng.integers(m); A[k, (j + 1) % m] = A[k, j]; b[k, (j + 1) % m] = b[k, j]
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    first = {k: v.clone() for k, v in pa.reduce_batch(At, bt).items()}
    diff = 0
    for r in range(reps):
        res = pa.reduce_batch(At, bt)
        for k in first:
            if not torch.equal(first[k].view(torch.uint8) if first[k].dtype != torch.float64 else first[k].view(torch.int64),
                               res[k].view(torch.uint8) if res[k].dtype != torch.float64 else res[k].view(torch.int64)):
                diff += 1
    bad += diff
    print("(%d,%d) B=%d: %d runs, %d differing arrays" % (m, d, B, reps, diff), flush=True)
print("SOAK", "CLEAN" if bad == 0 else "DIFFERENCES: %d" % bad)
