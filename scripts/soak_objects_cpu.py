#!/usr/bin/env python3
"""CPU soak of the HOST layer (polytope_amd/polytope.py, prop2partition.py) against the reference itself, both on the scipy
backend, in the build container only (/root/reference is imported in place; nothing of it is copied): random and structured
polytope pairs / small regions in d = 1..4 through reduce, intersect, union(check_convex), mldivide, envelope, is_convex,
is_adjacent, is_subset, ==, bounding_box, extreme, Region.intersect, Region diff, find_adjacent_regions -- same pieces in the
same order, rows within 1e-9, booleans equal.    python scripts/soak_objects_cpu.py [trials] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import logging  # noqa: E402
logging.disable(logging.CRITICAL)
import polytope as ref  # noqa: E402  (the reference, read in place)
import polytope_amd as mine  # noqa: E402
assert mine.solvers.default_solver != "hip"


def pieces(X):
    if isinstance(X, (ref.Region, mine.Region)):
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
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    nops = 0
    for trial in range(trials):
        d = int(rng.choice([1, 2, 2, 3, 3, 4]))
        kinds = [str(rng.choice(["box", "boxed", "boxed", "free"])) for _ in range(3)]
        data = [rand_poly(rng, d, k) for k in kinds]
        if trial % 4 == 0:      # a hyperplane split of the first polytope: touching pieces, convex union
            A0, b0 = data[0]
            n = rng.standard_normal(d); n /= np.linalg.norm(n)
            off = float(n @ rng.uniform(-0.2, 0.2, d))
            data[1] = (np.vstack([A0, n]), np.hstack([b0, off]))
            data[2] = (np.vstack([A0, -n]), np.hstack([b0, -off]))
        errs = []
        only = os.environ.get("SOAK_ONLY")
        if only is not None and int(only) != trial:
            continue

        def both(fn, what, cmp=same):
            nonlocal nops
            nops += 1
            out = []
            for mod in (ref, mine):
                t1 = time.time()
                P = [mod.Polytope(A.copy(), b.copy()) for A, b in data]
                try:
                    np.random.seed(trial)
                    out.append(("ok", fn(mod, P)))
                except Exception as e:   # the same exception class on both sides is parity too
                    out.append(("exc", type(e).__name__))
                if only is not None:
                    print("   %-22s %-10s %.2f s" % (what, mod.__name__, time.time() - t1), flush=True)
            if out[0][0] != out[1][0]:
                errs.append("%s: reference %s, package %s" % (what, out[0], out[1]))
            elif out[0][0] == "exc":
                if out[0][1] != out[1][1]:
                    errs.append("%s: exceptions %s / %s" % (what, out[0][1], out[1][1]))
            else:
                e = cmp(out[0][1], out[1][1], what)
                if e:
                    errs.append(e)

        eqb = lambda x, y, w: None if bool(x) == bool(y) else "%s: %s against %s" % (w, x, y)   # noqa: E731
        eqbox = lambda x, y, w: None if all(np.allclose(u, v, rtol=0, atol=1e-9, equal_nan=True) for u, v in zip(x, y)) else w + ": boxes differ"  # noqa: E731
        both(lambda m, P: m.reduce(P[0]), "reduce")
        both(lambda m, P: P[0].intersect(P[1]), "intersect")
        heavy = d <= 2   # (the reference's union(check_convex) chains cost minutes per call from d = 3 on: 2 ms per LP)
        if heavy:
            both(lambda m, P: m.union(P[1], P[2], check_convex=True), "union(check_convex)")
        both(lambda m, P: m.union(P[0], P[1], check_convex=False), "union")
        both(lambda m, P: m.mldivide(P[0], P[1]), "mldivide")
        both(lambda m, P: m.mldivide(P[0], m.Region([P[1], P[2]])), "mldivide(P, Region)")
        if heavy:
            both(lambda m, P: m.mldivide(m.Region([P[0], P[1]]), P[2]), "mldivide(Region, P)")
        both(lambda m, P: m.envelope(m.Region([P[1], P[2]])), "envelope")
        both(lambda m, P: m.is_convex(m.Region([P[1], P[2]]))[0], "is_convex", eqb)
        both(lambda m, P: m.is_adjacent(P[1], P[2]), "is_adjacent", eqb)
        both(lambda m, P: m.is_adjacent(P[0], P[1], overlap=True), "is_adjacent(overlap)", eqb)
        both(lambda m, P: m.is_subset(P[1], P[0]), "is_subset", eqb)
        if heavy:
            both(lambda m, P: m.is_subset(m.Region([P[1], P[2]]), P[0]), "is_subset(Region, P)", eqb)
        both(lambda m, P: P[1].bounding_box, "bounding_box", eqbox)
        both(lambda m, P: m.Region([P[0], P[1]]).bounding_box, "Region.bounding_box", eqbox)
        both(lambda m, P: m.is_fulldim(P[0].intersect(P[2])), "is_fulldim", eqb)
        if heavy:
            both(lambda m, P: m.Region([P[0], P[1]]).intersect(P[2]), "Region.intersect")
            both(lambda m, P: m.Region([P[1], P[2]]).diff(P[0]), "Region.diff")
        both(lambda m, P: m.cheby_ball(P[0])[0], "cheby_ball", lambda x, y, w: None if abs(x - y) <= 1e-9 else "%s: %r against %r" % (w, x, y))
        if d <= 3:
            both(lambda m, P: m.extreme(P[1]), "extreme",
                 lambda x, y, w: None if (x is None and y is None) or (x is not None and y is not None and x.shape == y.shape
                                                                        and np.allclose(np.sort(x, 0), np.sort(y, 0), atol=1e-7)) else w + ": vertices differ")
        if errs:
            bad += 1
            print("trial %d  d %d %s:" % (trial, d, kinds), "; ".join(errs[:4]), flush=True)
        elif os.environ.get("SOAK_VERBOSE"):
            print("trial %d  d %d %s ok  %.0f s" % (trial, d, kinds, time.time() - t0), flush=True)
    print("OBJECT SOAK (cpu, scipy backend on both sides) %s: %d trials, %d operations, %d trials with a difference, %.0f s" % (
        "FAILED" if bad else "OK", trials, nops, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
