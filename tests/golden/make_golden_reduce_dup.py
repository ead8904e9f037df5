#!/usr/bin/env python3
"""g23_reduce_dup.npz -- the reference's reduce() (polytope/polytope.py:1053-1163) on polytopes with rows a hair apart, on
elongated (rows not normalised) and on shifted ones, d = 2..8: the families on which the dictionary engines' absolute
tolerances (pivot 1e-7, reduced cost 1e-9: plp_common.hpp) were chosen, checked here against the REAL reference instead of
the oracle that shares them.  Which input rows the reference keeps, whether it calls the polytope empty,
its Chebyshev radius; run twice -- HiGHS at its defaults and with feasibility tolerances of 1e-10 -- and `pinned` where
both agree (the same rule as g17: a verdict that flips with the solver's tolerance is not the polytope's) AND no redundancy
LP of either run ended with a status other than optimal / unbounded: reduce() drops the row of such an LP (ref :1152-1160),
which on shifted polytopes is HiGHS's "numerical difficulties" (status 4), not a property of the row (`lp_trouble` counts them).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_reduce_dup.py      (build container only)"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, HERE)
import polytope as pc  # noqa: E402  (the reference)
import polytope.polytope as alg  # noqa: E402
from scipy.optimize import linprog  # noqa: E402
import soak_lane as SL  # noqa: E402
from make_golden_bbox_dup import make_one  # noqa: E402


def reference_reduce(A, b, tight):
    saved = alg.lpsolve
    trouble = [0]   # LPs of this call that HiGHS ended with a status other than optimal / unbounded (4: numerical difficulties,
                    # 2 from its presolve ...): reduce() then DROPS the row (ref :1152-1160) -- the solver's accident, not the polytope's

    def lp(c, G, h, solver=None):
        if tight:
            sol = linprog(c, G, np.transpose(h), None, None, bounds=(None, None),
                          options={"primal_feasibility_tolerance": 1e-10, "dual_feasibility_tolerance": 1e-10})
            r = dict(status=sol.status, x=sol.x, fun=sol.fun)
        else:
            r = saved(c, G, h, solver)
        if r["status"] not in (0, 3) and G.shape[1] == A.shape[1]:
            trouble[0] += 1
        return r
    alg.lpsolve = lp
    try:
        p = pc.Polytope(A.copy(), b.copy(), normalize=False)
        q = pc.reduce(p)
    finally:
        alg.lpsolve = saved
    m = A.shape[0]
    keep = np.zeros(m, bool)
    empty = q.A.size == 0
    rad = float(p._chebR) if p._chebR is not None else np.nan
    if not empty:
        scale = 1 / np.sqrt(np.sum(A * A, axis=1))
        An, bn = A * scale[:, None], b * scale
        used = []
        for a, bb in zip(q.A, q.b):   # (rows come back renormalised; of input rows on one plane the dedupe leaves the LAST of the smallest b)
            cand = np.nonzero((np.abs(An - a).max(1) < 1e-12) & (np.abs(bn - bb) < 1e-9))[0]
            cand = [c for c in cand if c not in used]
            assert len(cand), (a, bb)
            best = [c for c in cand if bn[c] == min(bn[cc] for cc in cand)][-1]
            used.append(best)
        keep[used] = True
    return keep, bool(empty), bool(q.minrep), rad, trouble[0]


def main():
    rng = np.random.default_rng(23)
    warnings.simplefilter("ignore")
    recs = []
    shapes = [(2, 8), (2, 14), (3, 10), (3, 16), (3, 28), (4, 12), (4, 24), (5, 16), (6, 20), (6, 32), (8, 24), (8, 40)]
    for rep in range(7):
        for (d, m) in shapes:
            for fam in ("dup", "dup", "long", "shift"):
                A, b = make_one(rng, fam, m, d)
                k0 = reference_reduce(A, b, False)
                k1 = reference_reduce(A, b, True)
                recs.append(dict(fam=fam, A=A, b=b, keep=k0[0], empty=k0[1], minrep=k0[2], r=k0[3], r_tight=k1[3],
                                 lp_trouble=k0[4] + k1[4],
                                 pinned=bool((k0[0] == k1[0]).all() and k0[1] == k1[1] and k0[2] == k1[2] and k0[4] + k1[4] == 0)))
    # polytopes the soaks found (tests/golden/found/*.npz: inputs only), family "found"
    import glob
    for f in sorted(glob.glob(os.path.join(HERE, "found", "*.npz"))):
        z = np.load(f)
        A, b = z["A"], z["b"]
        k0 = reference_reduce(A, b, False)
        k1 = reference_reduce(A, b, True)
        recs.append(dict(fam="found", A=A, b=b, keep=k0[0], empty=k0[1], minrep=k0[2], r=k0[3], r_tight=k1[3],
                         lp_trouble=k0[4] + k1[4],
                         pinned=bool((k0[0] == k1[0]).all() and k0[1] == k1[1] and k0[2] == k1[2] and k0[4] + k1[4] == 0)))
    mmax = max(r["A"].shape[0] for r in recs)
    dmax = max(r["A"].shape[1] for r in recs)

    def pad(rows, width, fill=np.nan):
        out = np.full((len(rows), width), fill)
        for i, r in enumerate(rows):
            out[i, :len(r)] = r
        return out
    out = dict(fam=np.array([r["fam"] for r in recs]), m=np.array([r["A"].shape[0] for r in recs]),
               d=np.array([r["A"].shape[1] for r in recs]), A=pad([r["A"].ravel() for r in recs], mmax * dmax),
               b=pad([r["b"] for r in recs], mmax), keep=pad([r["keep"].astype(float) for r in recs], mmax, 0.0).astype(bool),
               empty=np.array([r["empty"] for r in recs]), minrep=np.array([r["minrep"] for r in recs]),
               r=np.array([r["r"] for r in recs]), r_tight=np.array([r["r_tight"] for r in recs]),
               pinned=np.array([r["pinned"] for r in recs]), lp_trouble=np.array([r["lp_trouble"] for r in recs]))
    np.savez_compressed(os.path.join(HERE, "g23_reduce_dup.npz"), **out)
    pin = out["pinned"]
    print("g23: %d polytopes; empty %d; verdicts that depend on HiGHS's feasibility tolerance: %d (by family: %s)" % (
        len(recs), int(out["empty"].sum()), int((~pin).sum()),
        {f: int((~pin[out["fam"] == f]).sum()) for f in sorted(set(out["fam"]))}))


if __name__ == "__main__":
    main()
