#!/usr/bin/env python3
"""GPU box: the host layer on the 'hip' backend against the SAME layer on the 'scipy' backend -- which scripts/soak_objects_cpu.py
holds against the reference itself in the build container -- on random polytope triples in d = 1..4: reduce, intersect, union,
union(check_convex), mldivide (polytope / region forms), envelope, is_convex, is_adjacent, is_subset, bounding_box, Region.intersect,
Region.diff, cheby_ball, extreme.  Same pieces in the same order (rows 1e-9), same booleans.
Usage: gpurun --timeout 1500 -- 'python scripts/soak_objects_hip.py [trials] [seed]'"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import logging  # noqa: E402
logging.disable(logging.CRITICAL)
import polytope_amd as pc  # noqa: E402
from polytope_amd import solvers  # noqa: E402


def pieces(X):
    if isinstance(X, pc.Region):
        return list(X.list_poly)
    return [X] if X.A.size else []


def same(X, Y, what, tol=1e-9):
    a, b = pieces(X), pieces(Y)
    if len(a) != len(b):
        return "%s: %d pieces against %d" % (what, len(a), len(b))
    for k, (p, q) in enumerate(zip(a, b)):
        if p.A.shape != q.A.shape:
            return "%s: piece %d has %s rows against %s" % (what, k, p.A.shape, q.A.shape)
        if not (np.allclose(p.A, q.A, rtol=0, atol=tol) and np.allclose(p.b, q.b, rtol=0, atol=tol)):
            return "%s: piece %d rows differ by %.2e" % (what, k, max(np.abs(p.A - q.A).max(), np.abs(p.b - q.b).max()))
    return None


def rand_poly(rng, d, kind):
    if kind == "box":
        lo = rng.uniform(-1, 0.5, d)
        hi = lo + rng.uniform(0.3, 1.5, d)
        return np.vstack([np.eye(d), -np.eye(d)]), np.hstack([hi, -lo])
    m = int(rng.integers(d + 1, 4 * d + 3))
    A = rng.standard_normal((m, d))
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    c = rng.uniform(-0.4, 0.4, d)
    b = A @ c + rng.uniform(0.3, 1.0, m)
    if kind == "boxed":
        A = np.vstack([A, np.eye(d), -np.eye(d)])
        b = np.hstack([b, np.full(d, 1.5), np.full(d, 1.5)])
    return A, b


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert "hip" in solvers.installed_solvers, "needs the HIP backend"
    rng = np.random.default_rng(seed)
    bad = nops = 0
    unstable = [0]
    ties = [0]
    highs4 = [0]
    t0 = time.time()
    eqb = lambda x, y, w: None if bool(x) == bool(y) else "%s: %s against %s" % (w, x, y)   # noqa: E731
    eqbox = lambda x, y, w: None if all(np.allclose(u, v, rtol=0, atol=1e-9, equal_nan=True) for u, v in zip(x, y)) else w + ": boxes differ"  # noqa: E731
    for trial in range(trials):
        d = int(rng.choice([1, 2, 2, 3, 3, 4, 4]))
        kinds = [str(rng.choice(["box", "boxed", "boxed", "free"])) for _ in range(3)]
        data = [rand_poly(rng, d, k) for k in kinds]
        if trial % 4 == 0:
            A0, b0 = data[0]
            n = rng.standard_normal(d); n /= np.linalg.norm(n)
            off = float(n @ rng.uniform(-0.2, 0.2, d))
            data[1] = (np.vstack([A0, n]), np.hstack([b0, off]))
            data[2] = (np.vstack([A0, -n]), np.hstack([b0, -off]))
        errs = []

        def both(fn, what, cmp=same):
            nonlocal nops
            nops += 1
            out = []
            for backend in ("scipy", "hip"):
                solvers.default_solver = backend
                P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
                try:
                    np.random.seed(trial)
                    out.append(("ok", fn(P)))
                except Exception as e:
                    out.append(("exc", type(e).__name__))
            e = None
            if out[0][0] != out[1][0] or (out[0][0] == "exc" and out[0][1] != out[1][1]):
                e = "%s: scipy %s, hip %s" % (what, out[0] if out[0][0] == "exc" else "ok", out[1] if out[1][0] == "exc" else "ok")
            elif out[0][0] == "ok":
                e = cmp(out[0][1], out[1][1], what)
            if e:
                # Is the scipy side -- the reference's own flow -- reproducible on this input at all?  `==` between pieces of
                # tiny volume is decided by volumes sampled with an UNSEEDED generator (ref :1586, :220-230), and list.remove
                # acts on it (ref :1226-1228): two more runs of the same backend.
                again = []
                for _ in range(2):
                    solvers.default_solver = "scipy"
                    pc.polytope._hull_memo.clear(); pc.polytope._convex_memo.clear()
                    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
                    try:
                        again.append(("ok", fn(P)))
                    except Exception as ex:
                        again.append(("exc", type(ex).__name__))
                stable = all(a[0] == out[0][0] and (a[1] == out[0][1] if a[0] == "exc" else cmp(out[0][1], a[1], what) is None) for a in again)
                tie = False
                if stable and what == "mldivide(P, Region)":
                    # region_diff visits the cells in the order argsort(-Rc) (ref :2153-2157); two cells that both hold the minuend's
                    # own Chebyshev ball have the SAME radius, and which comes first in the reference is the last bit of its LP code
                    solvers.default_solver = "hip"
                    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
                    rc = [float(v) for v in pc.polytope._radii_stacked(P[0].copy(), [P[1], P[2]])]
                    tie = abs(rc[0] - rc[1]) <= 1e-12 * max(1.0, abs(rc[0])) and rc[0] > 0
                trouble = False
                if stable and not tie and out[1][0] == "ok":
                    # The scipy side RAISED where the hip side answered -- the reference raises RuntimeError when a bounding-box /
                    # Chebyshev LP comes back with a status other than optimal / infeasible / unbounded (ref :1378-1384) -- or it
                    # answered something else: reduce() DROPS the row whose redundancy LP ends that way (ref :1152-1160: seed 83,
                    # trial 19: a facet with a margin of 0.1 gone).  For HiGHS such a status is "numerical difficulties" (4) or a
                    # limit (1): the solver's accident on this LP, not a property of the polytope (g23: `lp_trouble`).  Looked for,
                    # not assumed: the scipy side once more with its LPs watched.
                    seen = []
                    orig = solvers.lpsolve

                    def watched(c, G, h, solver=None):
                        r = orig(c, G, h, solver)
                        if r["status"] not in (0, 2, 3):
                            seen.append(r["status"])
                        return r
                    solvers.lpsolve = watched
                    pc.polytope.lpsolve = watched      # (polytope.py binds the name at import as well)
                    solvers.default_solver = "scipy"
                    try:
                        fn([pc.Polytope(A.copy(), b.copy()) for A, b in data])
                    except Exception:
                        pass
                    finally:
                        solvers.lpsolve = orig
                        pc.polytope.lpsolve = orig
                    trouble = len(seen) > 0
                if not stable:
                    unstable[0] += 1
                elif tie:
                    ties[0] += 1
                elif trouble:
                    highs4[0] += 1
                    print("trial %d  %s: the scipy backend %s -- HiGHS ended an LP of it with status %s; the hip backend answers" % (
                        trial, what, "raises " + out[0][1] if out[0][0] == "exc" else "answers differently", sorted(set(seen))), flush=True)
                else:
                    errs.append(e)

        heavy = d <= 2 or trial % 5 == 0     # (the scipy side of the chains of union(check_convex) is what takes the time)
        both(lambda P: pc.reduce(P[0]), "reduce")
        both(lambda P: P[0].intersect(P[1]), "intersect")
        both(lambda P: pc.union(P[0], P[1], check_convex=False), "union")
        both(lambda P: pc.mldivide(P[0], P[1]), "mldivide")
        both(lambda P: pc.mldivide(P[0], pc.Region([P[1], P[2]])), "mldivide(P, Region)")
        both(lambda P: pc.envelope(pc.Region([P[1], P[2]])), "envelope")
        both(lambda P: pc.is_convex(pc.Region([P[1], P[2]]))[0], "is_convex", eqb)
        both(lambda P: pc.is_adjacent(P[1], P[2]), "is_adjacent", eqb)
        both(lambda P: pc.is_adjacent(P[0], P[1], overlap=True), "is_adjacent(overlap)", eqb)
        both(lambda P: pc.is_subset(P[1], P[0]), "is_subset", eqb)
        both(lambda P: P[1].bounding_box, "bounding_box", eqbox)
        both(lambda P: pc.Region([P[0], P[1]]).bounding_box, "Region.bounding_box", eqbox)
        both(lambda P: pc.is_fulldim(P[0].intersect(P[2])), "is_fulldim", eqb)
        both(lambda P: pc.cheby_ball(P[0])[0], "cheby_ball", lambda x, y, w: None if abs(x - y) <= 1e-9 else "%s: %r against %r" % (w, x, y))
        if heavy:
            both(lambda P: pc.union(P[1], P[2], check_convex=True), "union(check_convex)")
            both(lambda P: pc.mldivide(pc.Region([P[0], P[1]]), P[2]), "mldivide(Region, P)")
            both(lambda P: pc.Region([P[0], P[1]]).intersect(P[2]), "Region.intersect")
            both(lambda P: pc.Region([P[1], P[2]]).diff(P[0]), "Region.diff")
            both(lambda P: pc.is_subset(pc.Region([P[1], P[2]]), P[0]), "is_subset(Region, P)", eqb)
        if d <= 3:
            both(lambda P: pc.extreme(P[1]), "extreme",
                 lambda x, y, w: None if (x is None and y is None) or (x is not None and y is not None and x.shape == y.shape
                                                                        and np.allclose(np.sort(x, 0), np.sort(y, 0), atol=1e-7)) else w + ": vertices differ")
        if errs:
            bad += 1
            print("trial %d  d %d %s:" % (trial, d, kinds), "; ".join(errs[:4]), flush=True)
    solvers.default_solver = "scipy"
    print("OBJECT SOAK ('hip' against 'scipy' backend) %s: %d trials, %d operations, %d trials with a difference, %.0f s  (operations whose "
          "result the scipy backend itself does not reproduce -- sampled volumes deciding `==`: %d; region_diff on two cells of EQUAL radius, "
          "ordered by the last bit of the LP code: %d; operations on which the scipy backend raises or answers differently because HiGHS "
          "ended one of its LPs with status 1 / 4: %d)" % (
              "FAILED" if bad else "OK", trials, nops, bad, time.time() - t0, unstable[0], ties[0], highs4[0]), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
