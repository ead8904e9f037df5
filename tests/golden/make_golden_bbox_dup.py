#!/usr/bin/env python3
"""g22_bbox_dup.npz -- the reference's bounding_box (polytope/polytope.py:1314-1411) and cheby_ball (:1241-1300) on
polytopes with rows a hair apart, slivers and elongated shapes, d = 2..16: the inputs bounding_box meets when it is handed
an un-reduced stack (it has no dedupe in front of its 2d LPs), and where round 5's kernels were off by up to 7e-6.

Run in the build container only (imports the reference from /root/reference, scipy / HiGHS backend):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bbox_dup.py

Per polytope the fixture holds the rows, the REFERENCE's box / Chebyshev radius (or the fact that it raised), and the
ORACLE's (oracle/plp_oracle.c: certified against the original rows, binary128 where the double engine's answer does not
certify) with `dev_*` = how far the reference is from it, relative to the box's extent.  HiGHS works to a primal / dual
feasibility tolerance of 1e-7, so on these inputs the reference itself is 1e-9 .. 1e-7 (of the extent) away from the optimum
of the LP as given on a part of the cases -- the fixture says so per case instead of a loosened tolerance for all:
a test holds the kernels to 1e-9 of the oracle everywhere, and to 1e-9 of the reference wherever dev <= 1e-9.
Families: `dup` (copies of rows 1e-16 .. 1e-5 rad away, right-hand sides equal or 1e-7 / 0.1 apart: scripts/soak_lane.py),
`sliver` (two rows 1e-11 .. 1e-6 rad apart FACING each other: extents of 1e5 .. 1e16), `long` (coordinates scaled by up to 1e4:
elongated polytopes, rows not normalised), `shift` (dup, translated by up to 1e3: nothing is near the origin)."""
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
import polytope as pc  # noqa: E402  (the reference)
from polytope import solvers  # noqa: E402
import soak_lane as SL  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert solvers.default_solver == "scipy", solvers.default_solver


def make_one(rng, fam, m, d):
    if fam in ("dup", "shift"):
        A, b, _ = SL.make(rng, 1, m, d, "dup")
        A, b = A[0], b[0]
        if fam == "shift":
            b = b + A @ (10.0 ** rng.uniform(0, 3) * rng.standard_normal(d))
    elif fam == "sliver":
        A, b, _ = SL.make(rng, 1, m, d, "random")
        A, b = A[0].copy(), b[0].copy()
        i, j = rng.choice(m, 2, replace=False)
        eps = 10.0 ** rng.uniform(-11, -6)
        t = rng.standard_normal(d)
        A[j] = -A[i] + eps * t          # faces row i, a hair from antiparallel
        A[j] /= np.linalg.norm(A[j])
        b[j] = -b[i] + 10.0 ** rng.uniform(-6, -1)
        if m >= 2 * d:                   # and it replaces a box row: the polytope may reach far along the hair
            k = int(rng.integers(0, 2 * d))
            A[k], b[k] = A[j], b[j]
    else:  # long
        A, b, _ = SL.make(rng, 1, m, d, "random")
        s = 10.0 ** rng.uniform(-2, 2, d)
        A, b = A[0] / s[None, :], b[0].copy()   # x_k stretched by s_k; rows left un-normalised
    return np.ascontiguousarray(A), np.ascontiguousarray(b)


def ext_of(lo, hi):
    v = np.concatenate([lo, hi])
    v = v[np.isfinite(v)]
    return max(1.0, float(np.max(np.abs(v), initial=1.0)))


def dev(ref, ora, ext):
    """largest distance of the reference's finite sides from the oracle's, relative to the extent; inf where +-inf / nan
    sit in different places"""
    fr, fo = np.isfinite(ref), np.isfinite(ora)
    if not np.array_equal(fr, fo) or not np.array_equal(ref[~fr], ora[~fo]):
        return np.inf
    return float(np.max(np.abs(ref[fr] - ora[fo]), initial=0.0)) / ext


def main():
    rng = np.random.default_rng(22)
    recs = []
    shapes = [(2, 8), (2, 12), (3, 10), (3, 16), (3, 28), (4, 12), (4, 24), (5, 14), (6, 20), (6, 32), (8, 20), (8, 40),
              (10, 24), (13, 30), (16, 36)]
    fams = ["dup", "dup", "sliver", "long", "shift"]
    warnings.simplefilter("ignore")
    for rep in range(9):
        for (d, m) in shapes:
            for fam in fams:
                if len(recs) >= 640:
                    break
                A, b = make_one(rng, fam, m, d)
                P = pc.Polytope(A.copy(), b.copy(), normalize=False)
                try:
                    lo, hi = pc.bounding_box(P)
                    rlo, rhi, rerr = np.asarray(lo, float).ravel(), np.asarray(hi, float).ravel(), 0
                except RuntimeError:
                    rlo, rhi, rerr = np.full(d, np.nan), np.full(d, np.nan), 1
                P2 = pc.Polytope(A.copy(), b.copy(), normalize=False)
                rr, _ = pc.cheby_ball(P2)
                olo, ohi, obad = O.bounding_box(A, b)
                ost, orr, _ = O.cheby(A, b)
                orr = orr if (ost == 0 and orr >= 0) else 0.0
                e = ext_of(olo, ohi)
                recs.append(dict(fam=fam, A=A, b=b, rlo=rlo, rhi=rhi, rerr=rerr, rr=float(rr), olo=olo, ohi=ohi, obad=int(obad),
                                 orr=float(orr), ext=e,
                                 dev_box=(np.inf if rerr else dev(np.concatenate([rlo, rhi]), np.concatenate([olo, ohi]), e)),
                                 dev_r=abs(float(rr) - float(orr)) / max(1.0, abs(float(orr)))))
    mmax = max(r["A"].shape[0] for r in recs)
    dmax = max(r["A"].shape[1] for r in recs)

    def pad(rows, width):
        out = np.full((len(rows), width), np.nan)
        for i, r in enumerate(rows):
            out[i, :len(r)] = r
        return out
    out = dict(fam=np.array([r["fam"] for r in recs]), m=np.array([r["A"].shape[0] for r in recs]),
               d=np.array([r["A"].shape[1] for r in recs]), A=pad([r["A"].ravel() for r in recs], mmax * dmax),
               b=pad([r["b"] for r in recs], mmax), ref_lb=pad([r["rlo"] for r in recs], dmax),
               ref_ub=pad([r["rhi"] for r in recs], dmax), ref_err=np.array([r["rerr"] for r in recs]),
               ref_r=np.array([r["rr"] for r in recs]), ora_lb=pad([r["olo"] for r in recs], dmax),
               ora_ub=pad([r["ohi"] for r in recs], dmax), ora_bad=np.array([r["obad"] for r in recs]),
               ora_r=np.array([r["orr"] for r in recs]), ext=np.array([r["ext"] for r in recs]),
               dev_box=np.array([r["dev_box"] for r in recs]), dev_r=np.array([r["dev_r"] for r in recs]))
    np.savez_compressed(os.path.join(HERE, "g22_bbox_dup.npz"), **out)
    db, dr = out["dev_box"], out["dev_r"]
    print("g22: %d polytopes (d = 2..16); reference raised on %d; its box is within 1e-9 of the oracle's on %d, 1e-9..1e-7 on %d, "
          "beyond (or +-inf elsewhere) on %d; radius within 1e-9 on %d, beyond on %d (largest %.1e)" % (
              len(recs), int(out["ref_err"].sum()), int((db <= 1e-9).sum()), int(((db > 1e-9) & (db <= 1e-7)).sum()),
              int((db > 1e-7).sum()), int((dr <= 1e-9).sum()), int((dr > 1e-9).sum()), float(dr.max())))
    for f in sorted(set(out["fam"])):
        sel = out["fam"] == f
        print("   %-7s n %3d  box dev <= 1e-9: %3d   <= 1e-7: %3d   beyond: %3d   infinite sides in %d" % (
            f, sel.sum(), (db[sel] <= 1e-9).sum(), ((db[sel] > 1e-9) & (db[sel] <= 1e-7)).sum(), (db[sel] > 1e-7).sum(),
            int(sum(np.isinf(out["ora_lb"][i, :out["d"][i]]).any() or np.isinf(out["ora_ub"][i, :out["d"][i]]).any() for i in np.nonzero(sel)[0]))))


if __name__ == "__main__":
    main()
