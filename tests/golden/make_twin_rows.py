"""tests/golden/twin_rows.npz: three polytopes of the soak scripts' `dup` family (rows a hair apart) on which a dictionary simplex
that accepts pivots down to 1e-9 ends with a Chebyshev ball sticking 0.05 .. 6 out of the polytope, or calls the ball LP unbounded
-- with the radius scipy / HiGHS (the reference's backend, solvers.py:152-154) finds for them.
    python tests/golden/make_twin_rows.py      (regenerates the inputs from the soak scripts' RNG streams)"""
import os, sys
import numpy as np
from scipy.optimize import linprog
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL

FAMS = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]


def lane_batch(seed, want):
    rng = np.random.default_rng(seed)
    for trial in range(want + 1):
        d = int(rng.choice([1, 2, 3, 3, 3, 4, 4])); m = int(rng.integers(d + 1, 33)); cls = trial % 6
        B = [int(rng.integers(1, 300)), int(rng.integers(2000, 9000)), int(rng.integers(15000, 30000)),
             int(rng.integers(41000, 60000)), int(rng.integers(300, 2000)), int(rng.integers(30001, 36000))][cls]
        if m > 16 or d == 4:
            B = min(B, 36000)
        fam = FAMS[int(rng.integers(0, len(FAMS)))]
        if rng.random() < 0.5:
            if rng.random() < 0.4:
                rng.choice([4, 8, 16])
        A, b, mrows = SL.make(rng, B, m, d, fam)
    return A, b


def wide_batch(seed, want):
    rng = np.random.default_rng(seed)
    for trial in range(want + 1):
        d = int(rng.choice([4, 5, 5, 6, 6, 7, 8, 8, 9, 10, 12, 13, 14, 16])); m = int(rng.integers(d + 1, 65)); cls = trial % 5
        B = [int(rng.integers(1, 200)), int(rng.integers(1000, 3000)), int(rng.integers(4000, 9000)),
             int(rng.integers(12000, 22000)), int(rng.integers(300, 1000))][cls]
        if d >= 12 or m > 48:
            B = min(B, 6000)
        fam = FAMS[int(rng.integers(0, len(FAMS)))]
        A, b, mrows = SL.make(rng, B, m, d, fam)
    return A, b


def highs_r(A, b):
    nrm = np.sqrt(np.sum(A * A, 1))
    c = np.zeros(A.shape[1] + 1); c[-1] = -1.0
    rs = linprog(c, np.hstack([A, nrm[:, None]]), b, bounds=(None, None))
    assert rs.status == 0
    return -rs.fun


out = {}
A, b = lane_batch(103, 25)
for name, k in (("a", 2546), ("b", 1853)):
    out["A_" + name], out["b_" + name], out["r_" + name] = A[k], b[k], highs_r(A[k], b[k])
A, b = wide_batch(1, 33)
out["A_c"], out["b_c"], out["r_c"] = A[1699], b[1699], highs_r(A[1699], b[1699])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "twin_rows.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})
