#!/usr/bin/env python3
"""CPU: EDGE inputs through the host layer against the reference itself (both on the scipy backend; build container only, the
reference is imported in place): empty polytopes, half-spaces, slabs, flat (lower-dimensional) sets, open cones, tiny boxes,
duplicated and zero rows, one-member and empty Regions -- every pair of a menu through the light operations; results (pieces in
order, rows 1e-9, booleans, boxes) or the exception CLASS must agree.      python scripts/soak_edges_cpu.py [d ...]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, "/root/reference")
import logging  # noqa: E402
logging.disable(logging.CRITICAL)
import polytope as ref  # noqa: E402
import polytope_amd as mine  # noqa: E402
from soak_objects_cpu import same  # noqa: E402


def menu(d):
    I = np.eye(d)
    box = (np.vstack([I, -I]), np.r_[np.ones(d), np.zeros(d)])
    items = {
        "empty": (np.zeros((0, d)), np.zeros(0)),
        "box": box,
        "box_shift": (box[0], box[1] + box[0] @ (0.5 * np.ones(d))),
        "box_touch": (box[0], box[1] + box[0] @ np.r_[1.0, np.zeros(d - 1)]),
        "box_apart": (box[0], box[1] + box[0] @ (3.0 * np.ones(d))),
        "tiny": (box[0], 1e-3 * box[1]),
        "halfspace": (I[:1], np.array([0.5])),
        "slab": (np.vstack([I[:1], -I[:1]]), np.array([0.7, -0.3])),
        "flat": (np.vstack([box[0], I[:1], -I[:1]]), np.r_[box[1], 0.5, -0.5]),
        "infeasible": (np.vstack([box[0], I[:1]]), np.r_[box[1], -1.0]),
        "cone": (-I, np.zeros(d)),
        "dup_rows": (np.vstack([box[0], box[0][:2]]), np.r_[box[1], box[1][:2] + [0.0, 0.2]]),
        "zero_row": (np.vstack([box[0], np.zeros((1, d))]), np.r_[box[1], 1.0]),
        "simplex": (np.vstack([-I, np.ones((1, d)) / np.sqrt(d)]), np.r_[np.zeros(d), 0.6]),
    }
    return items


def main():
    dims = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
    bad = nops = 0
    t0 = time.time()
    eqb = lambda x, y, w: None if bool(x) == bool(y) else "%s: %s against %s" % (w, x, y)   # noqa: E731

    def eqbox(x, y, w):
        ok = all(np.allclose(np.asarray(u, float), np.asarray(v, float), rtol=0, atol=1e-9, equal_nan=True) for u, v in zip(x, y))
        return None if ok else "%s: %s against %s" % (w, [np.ravel(u) for u in x], [np.ravel(v) for v in y])

    def eqnum(x, y, w):
        return None if (x is None and y is None) or abs(float(x) - float(y)) <= 1e-9 else "%s: %r against %r" % (w, x, y)

    for d in dims:
        items = menu(d)
        ops1 = [
            ("reduce", lambda m, P: m.reduce(P), same),
            ("cheby_ball", lambda m, P: m.cheby_ball(P)[0], eqnum),
            ("is_fulldim", lambda m, P: m.is_fulldim(P), eqb),
            ("is_empty", lambda m, P: m.is_empty(P), eqb),
            ("bounding_box", lambda m, P: P.bounding_box, eqbox),
            ("Region([P])", lambda m, P: m.Region([P]), same),
            ("Region([P, empty])", lambda m, P: m.Region([P, m.Polytope()]), same),
            ("Region bounding_box", lambda m, P: m.Region([P, P.copy()]).bounding_box, eqbox),
            ("copy ==", lambda m, P: P == P.copy(), eqb),
            ("contains centre", lambda m, P: (0.5 * np.ones((d, 1))) in P if P.A.size else False, eqb),
            ("dim", lambda m, P: P.dim, lambda x, y, w: None if x == y else w + ": %r against %r" % (x, y)),
        ]
        ops2 = [
            ("intersect", lambda m, P, Q: P.intersect(Q), same),
            ("is_adjacent", lambda m, P, Q: m.is_adjacent(P, Q), eqb),
            ("is_subset", lambda m, P, Q: m.is_subset(P, Q), eqb),
            ("mldivide", lambda m, P, Q: m.mldivide(P, Q), same),
            ("union", lambda m, P, Q: m.union(P, Q), same),
            ("envelope", lambda m, P, Q: m.envelope(m.Region([P, Q])), same),
            ("is_convex", lambda m, P, Q: m.is_convex(m.Region([P, Q]))[0], eqb),
            ("<=", lambda m, P, Q: P <= Q, eqb),
        ]
        if d <= 2:
            ops2.append(("union(check_convex)", lambda m, P, Q: m.union(P, Q, check_convex=True), same))

        def run(what, fn, cmp, names):
            nonlocal bad, nops
            nops += 1
            out = []
            for mod in (ref, mine):
                args = [mod.Polytope(items[n][0].copy(), items[n][1].copy()) for n in names]
                try:
                    np.random.seed(0)
                    out.append(("ok", fn(mod, *args)))
                except Exception as e:
                    out.append(("exc", type(e).__name__))
            err = None
            if out[0][0] != out[1][0] or (out[0][0] == "exc" and out[0][1] != out[1][1]):
                err = "reference %s, package %s" % (out[0] if out[0][0] == "exc" else "ok", out[1] if out[1][0] == "exc" else "ok")
            elif out[0][0] == "ok":
                try:
                    err = cmp(out[0][1], out[1][1], what)
                except Exception as e:
                    err = "comparison failed: %r" % (e,)
            if err:
                bad += 1
                print("d=%d %-20s %-24s %s" % (d, what, "/".join(names), err), flush=True)

        for n in items:
            for what, fn, cmp in ops1:
                run(what, fn, cmp, [n])
        for n1, n2 in itertools.product(items, items):
            for what, fn, cmp in ops2:
                run(what, fn, cmp, [n1, n2])
        print("d=%d done: %d operations so far, %d differences, %.0f s" % (d, nops, bad, time.time() - t0), flush=True)
    print("EDGE SOAK (cpu, scipy backend on both sides) %s: %d operations, %d differences" % ("FAILED" if bad else "OK", nops, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
