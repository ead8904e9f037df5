#!/usr/bin/env python3
"""g24_small_entries.npz -- what the reference's LP solver does with TINY matrix entries, measured through the reference's own
`lpsolve` (polytope/solvers.py:76-106, 149-158; scipy.optimize.linprog -> HiGHS).  HiGHS takes a matrix entry of magnitude
<= 1e-9 for zero (its `small_matrix_value`), so for the reference a row tilted by 1e-16 .. 1e-9 from a twin IS that twin.  The
library's verifier / careful engine and the certified oracle read LPs the same way (csrc/plp_verify.hpp: LpView::g;
oracle/plp_oracle.c: plpo_lp_solve); this fixture pins the rule to the reference's answers instead of to a remembered constant.

Two kinds of LPs, each for a ladder of eps around 1e-9 (both signs) and L = 1e3, 1e6:
  `lever`   min x0  s.t.  -x0 + eps x1 <= 2,  |x1| <= L             kept: -2 - |eps| L       dropped: -2
  `box`     bounding box (F3, :1367-1396) of { -x0 + eps x1 <= 2, x0 <= 3, |x1| <= L }: the reference's own lb[0] / ub[0]
            (G, h hold the rows, c = (lb[0], ub[0]))

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_small_entries.py      (build container only)"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import polytope as pc  # noqa: E402  (the reference)
from polytope import solvers  # noqa: E402

assert solvers.default_solver == "scipy", solvers.default_solver


def main():
    warnings.simplefilter("ignore")
    eps_ladder = [0.0, 1e-16, 1e-12, 1e-10, 5e-10, 9e-10, 9.999999e-10, 1e-9, 1.0000001e-9, 1.1e-9, 2e-9, 1e-8, 1e-7, 1e-6]
    kinds, epss, Ls, Gs, hs, cs, stat, fun = [], [], [], [], [], [], [], []
    for L in (1e3, 1e6):
        for eps in eps_ladder:
            for sgn in (1.0, -1.0):
                e = sgn * eps
                # lever
                G = np.array([[-1.0, e], [0.0, 1.0], [0.0, -1.0]])
                h = np.array([2.0, L, L])
                c = np.array([1.0, 0.0])
                r = solvers.lpsolve(c, G, h)
                # (stored with a vacuous fourth row 0 <= 1 so that both kinds have four rows)
                kinds.append("lever"); epss.append(e); Ls.append(L); Gs.append(np.vstack([G, [[0.0, 0.0]]])); hs.append(np.r_[h, 1.0]); cs.append(c)
                stat.append(int(r["status"])); fun.append(float(r["fun"]) if r["status"] == 0 else np.nan)
                # box: lower corner of x0
                A = np.array([[-1.0, e], [1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
                b = np.array([2.0, 3.0, L, L])
                lb, ub = pc.Polytope(A, b, normalize=False).bounding_box
                kinds.append("box"); epss.append(e); Ls.append(L)
                Gs.append(A); hs.append(b); cs.append(np.array([float(lb[0, 0]), float(ub[0, 0])]))
                stat.append(0); fun.append(float(lb[0, 0]))
    out = dict(kind=np.array(kinds), eps=np.array(epss), L=np.array(Ls), G=np.stack(Gs), h=np.stack(hs), c=np.stack(cs),
               status=np.array(stat), fun=np.array(fun))
    np.savez_compressed(os.path.join(HERE, "g24_small_entries.npz"), **out)
    lev = out["kind"] == "lever"
    kept = np.abs(out["fun"][lev] - (-2.0 - np.abs(out["eps"][lev]) * out["L"][lev])) <= 1e-9 * np.maximum(1.0, np.abs(out["eps"][lev]) * out["L"][lev])
    drop = np.abs(out["fun"][lev] + 2.0) <= 1e-12
    ae = np.abs(out["eps"][lev])
    print("g24: %d LPs; lever: the entry is DROPPED (answer -2) for |eps| <= %.9g, KEPT (answer -2 - |eps| L) from %.9g on" % (
        len(kinds), ae[drop & (ae > 0)].max(), ae[kept & ~drop].min()))


if __name__ == "__main__":
    main()
