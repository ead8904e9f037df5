#!/usr/bin/env python3
"""Round-2 soak (GPU box): the new host/device paths against their slower twins on many random inputs.
  * region_diff: library search (plp_region_diff_search) == host loop over batched calls (pieces, order, rows)
  * quickhull:   native main loop (plp_quickhull_run) == Python facet graph (rows bitwise) == scipy ConvexHull vertices
  * LDS engine / one-LP-per-wavefront engine == lane-group engines == oracle, degenerate LPs included
Usage: gpurun --timeout 2400 -- 'python scripts/soak2.py [trials]'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from scipy.spatial import ConvexHull  # noqa: E402
import polytope_amd as pa  # noqa: E402
import polytope_amd.polytope as pc  # noqa: E402
import polytope_amd.quickhull as Q  # noqa: E402
from polytope_amd import solvers  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
solvers.default_solver = "hip"
TR = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
t0 = time.time()

# ---------------------------------------------------------------- region_diff
rng = np.random.default_rng(2025)
n_idx = n_ok = 0
for trial in range(TR):
    d = int(rng.integers(2, 6))
    n = int(rng.integers(2, 30 if d <= 3 else 14))
    kind = trial % 3
    if kind == 0:      # overlapping random boxes
        cen = rng.random((n, d)); hw = rng.uniform(0.05, 0.35, (n, d))
        cells = [pc.box2poly(np.c_[c - w, c + w].tolist()) for c, w in zip(cen, hw)]
    elif kind == 1:    # grid cells (ties, shared facets)
        shape = tuple(int(v) for v in rng.integers(1, 4, d))
        import itertools
        cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(d)]) for i in itertools.product(*[range(s) for s in shape])]
        cells = [cells[i] for i in rng.permutation(len(cells))[: max(1, len(cells) * 2 // 3)]]
    else:              # random polytopes
        cells = []
        for _ in range(n):
            A = rng.standard_normal((2 * d + 2, d)); A /= np.linalg.norm(A, axis=1)[:, None]
            c = rng.random(d)
            cells.append(pc.Polytope(A, rng.uniform(0.1, 0.4, 2 * d + 2) + A @ c))
    A = rng.standard_normal((3 * d, d)); A /= np.linalg.norm(A, axis=1)[:, None]
    P = pc.Polytope(A, rng.uniform(0.2, 0.6) * (1 + rng.random(3 * d)) + A @ (0.5 * np.ones(d)))
    out = []
    for native in (True, False):
        pc._RDIFF_NATIVE = native
        try:
            D = pc.region_diff(P.copy(), pc.Region([c.copy() for c in cells]))
            ps = list(D.list_poly) if isinstance(D, pc.Region) else ([] if D.A.size == 0 else [D])
            out.append([(q.A.copy(), q.b.copy()) for q in ps])
        except IndexError:
            out.append("IndexError")
    same = type(out[0]) is type(out[1])
    if same and out[0] != "IndexError":
        same = len(out[0]) == len(out[1]) and all(a[0].shape == c[0].shape and np.allclose(a[0], c[0], atol=1e-12, rtol=0)
                                                  and np.allclose(a[1], c[1], atol=1e-12, rtol=0) for a, c in zip(*out))
        n_ok += 1
    else:
        n_idx += 1
    if not same:
        bad += 1
        print("region_diff MISMATCH trial", trial, "d", d, "cells", len(cells), flush=True)
pc._RDIFF_NATIVE = True
print("region_diff: %d inputs (%d compared piece by piece, %d IndexError on both), mismatches so far %d, %.0f s" % (TR, n_ok, n_idx, bad, time.time() - t0), flush=True)

# ---------------------------------------------------------------- quickhull
rng = np.random.default_rng(77)
nb = 0
near = 0
own_diff = 0
worst_viol = 0.0
for trial in range(TR):
    d = int(rng.integers(2, 6))
    N = int(rng.integers(d + 2, 60000 if d < 4 else (6000 if d == 4 else 600)))
    P = rng.standard_normal((N, d)) if trial % 2 else rng.random((N, d))
    if trial % 5 == 0:
        P[N // 2:] = P[:N - N // 2]                 # duplicated points
    if trial % 7 == 0:
        P = np.round(P * 4) / 4                     # lattice: many coplanar points
    res = {}
    err = {}
    # three runs: the native loop with LAPACK's solves for every size (the Python graph's arithmetic: rows must agree
    # bitwise), the Python graph, and the native loop as shipped ("own": its own LU from 4096 points on, which may differ
    # from LAPACK's in the last bits of a facet)
    for native in (True, False, "own"):
        Q._NATIVE_LOOP = bool(native)
        if native is True:
            os.environ["PLP_QH_LAPACK_BELOW"] = str(1 << 60)
        else:
            os.environ.pop("PLP_QH_LAPACK_BELOW", None)
        np.random.seed(trial)
        try:
            res[native] = Q.quickhull(P)
        except Exception as e:  # degenerate input: all must fail alike
            err[native] = type(e).__name__
    if err:
        if not (err.get(True) == err.get(False) == err.get("own")):
            nb += 1
            print("quickhull error mismatch", trial, err, flush=True)
        continue
    A3, b3, V3 = res["own"]
    if not (A3.shape == res[False][0].shape and np.array_equal(A3, res[False][0]) and np.array_equal(b3, res[False][1])):
        # last-bit differences may flip a tie: then the two hulls must still be the same body
        own_diff += 1
        A2_, b2_, V2_ = res[False]
        if A3.size and A2_.size:
            same = (set(map(tuple, np.unique(V3, axis=0))) == set(map(tuple, np.unique(V2_, axis=0)))
                    or (float(np.max(A3 @ V2_.T - b3[:, None])) < 1e-6 and float(np.max(A2_ @ V3.T - b2_[:, None])) < 1e-6))
        else:
            same = A3.size == A2_.size
        if not same:
            nb += 1
            print("quickhull OWN-LU MISMATCH", trial, d, N, flush=True)
            continue
    (A1, b1, V1), (A2, b2, V2) = res[True], res[False]
    if A1.size == 0 or A2.size == 0:
        ok = A1.size == A2.size
    else:
        ok = A1.shape == A2.shape and np.array_equal(A1, A2) and np.array_equal(b1, b2) and np.array_equal(V1, V2)
        if ok and trial % 7 != 0 and trial % 5 != 0:
            # against qhull: every point within a few abs_tol of our hull (the reference's algorithm never re-tests a
            # point that was within abs_tol of the facets of its time, so it may miss a vertex that pokes out by ~1e-7),
            # and our vertices a subset of qhull's
            ref = np.unique(P[np.unique(ConvexHull(P).vertices)], axis=0)
            Vu = np.unique(V1, axis=0)
            refset = set(map(tuple, ref))
            viol = float(np.max(A1 @ P.T - b1[:, None]))
            worst_viol = max(worst_viol, viol)
            ok = viol < 1e-4     # sliver facets through nearly dependent vertices (the reference's own construction) tilt by ~1e-5
            ch = ConvexHull(Vu) if Vu.shape[0] > d + 1 else None
            extra = [v for v in Vu if tuple(v) not in refset]          # vertices qhull merged away (coplanar within its tolerance)
            if extra and ch is not None:
                eq = ConvexHull(ref).equations
                ok = ok and float(np.max(eq[:, :-1] @ np.array(extra).T + eq[:, -1:])) < 5e-7
            near += Vu.shape != ref.shape or bool(extra)
    if not ok:
        nb += 1
        print("quickhull MISMATCH", trial, d, N, flush=True)
Q._NATIVE_LOOP = True
bad += nb
print("quickhull: %d inputs, mismatches %d (vertex sets that differ from qhull's by points within 5e-7 of either hull: %d; largest point-facet violation %.1e; hulls where the shipped LU differs from LAPACK in some bit: %d, all the same body), %.0f s" % (TR, nb, near, worst_viol, own_diff, time.time() - t0), flush=True)

# ---------------------------------------------------------------- LP engines
from degenerate_cases import degenerate_lps  # noqa: E402
rng = np.random.default_rng(5)
ne = 0
for trial in range(TR):
    m = int(rng.integers(1, 65)); n = int(rng.integers(1, 18)); B = 24
    G = rng.standard_normal((B, m, n)); G /= np.linalg.norm(G, axis=2, keepdims=True)
    h = rng.random((B, m)) + 0.2
    h[::3] -= 0.6 * rng.random((len(h[::3]), m))
    if m >= 2 * n:
        G[::2, :2 * n] = np.vstack([np.eye(n), -np.eye(n)])[None]; h[::2, :2 * n] = 3.0
    if trial % 4 == 0 and m >= 2:
        G[:, m // 2:] = G[:, : m - m // 2]; h[:, m // 2:] = h[:, : m - m // 2]      # duplicated rows
    c = rng.standard_normal((B, n))
    ms = rng.integers(max(1, m - 3), m + 1, B).astype(np.int32)
    os.environ["PLP_LP_1ROW"] = "1"; ref = pa.lpsolve_batch(c, G, h, m=ms); del os.environ["PLP_LP_1ROW"]
    os.environ["PLP_LDS"] = "1"; got = pa.lpsolve_batch(c, G, h, m=ms); del os.environ["PLP_LDS"]
    dflt = pa.lpsolve_batch(c, G, h, m=ms)
    ok = np.array_equal(got["status"], ref["status"]) and np.array_equal(got["fun"], ref["fun"], equal_nan=True) \
        and np.array_equal(dflt["status"], ref["status"]) and np.allclose(dflt["fun"], ref["fun"], rtol=0, atol=1e-11, equal_nan=True)
    if n >= 2:
        d = n - 1
        A = np.ascontiguousarray(G[:, :, :d])
        os.environ["PLP_CHEBY_WIDE"] = "0"; r0 = pa.cheby_ball_batch(A, h, m=ms)
        os.environ["PLP_CHEBY_WIDE"] = "1"; r1 = pa.cheby_ball_batch(A, h, m=ms); del os.environ["PLP_CHEBY_WIDE"]
        os.environ["PLP_LDS"] = "1"; r2 = pa.cheby_ball_batch(A, h, m=ms); del os.environ["PLP_LDS"]
        ok = ok and np.array_equal(r0["status"], r1["status"]) and np.array_equal(r0["status"], r2["status"])
        good = r0["status"] == 0
        ok = ok and np.allclose(r0["r"][good], r1["r"][good], rtol=0, atol=1e-11) and np.allclose(r0["r"][good], r2["r"][good], rtol=0, atol=1e-11)
        for k in range(0, B, 8):
            so, ro, _ = O.cheby(A[k, :ms[k]], h[k, :ms[k]])
            ok = ok and so == r1["status"][k] and (so != 0 or abs(ro - r1["r"][k]) <= 1e-9 * max(1, abs(ro)))
    if not ok:
        ne += 1
        print("LP engines MISMATCH trial", trial, m, n, flush=True)
for kind, c, G, h in degenerate_lps(seed=9, reps=20):
    os.environ["PLP_LP_1ROW"] = "1"; ref = pa.lpsolve_batch(c[None], G[None], h[None]); del os.environ["PLP_LP_1ROW"]
    os.environ["PLP_LDS"] = "1"; got = pa.lpsolve_batch(c[None], G[None], h[None]); del os.environ["PLP_LDS"]
    if got["status"][0] != ref["status"][0] or not np.array_equal(got["fun"], ref["fun"], equal_nan=True):
        ne += 1
        print("degenerate LP MISMATCH", kind, flush=True)
    dflt = pa.lpsolve_batch(c[None], G[None], h[None])   # (n = 5..16: lp_w_kernel, Bland's rule inside the loop)
    if dflt["status"][0] != ref["status"][0] or not np.allclose(dflt["fun"], ref["fun"], rtol=0, atol=1e-11, equal_nan=True) \
            or (5 <= G.shape[1] <= 16 and not np.array_equal(dflt["iters"], ref["iters"])):
        ne += 1
        print("degenerate LP MISMATCH (default route)", kind, G.shape, dflt["status"], ref["status"], flush=True)
    if kind.endswith("F1") and G.shape[1] >= 6:
        A = np.ascontiguousarray(G[None, :, :-1]); 
        os.environ["PLP_CHEBY_WIDE"] = "0"; r0 = pa.cheby_ball_batch(A, h[None])
        os.environ["PLP_CHEBY_WIDE"] = "1"; r1 = pa.cheby_ball_batch(A, h[None]); del os.environ["PLP_CHEBY_WIDE"]
        if r0["status"][0] != r1["status"][0] or (r0["status"][0] == 0 and abs(r0["r"][0] - r1["r"][0]) > 1e-11):
            ne += 1
            print("degenerate F1 wide MISMATCH", kind, flush=True)
bad += ne
print("LP engines: %d shapes + degenerate set, mismatches %d, %.0f s" % (TR, ne, time.time() - t0), flush=True)
print("SOAK2", "FAILED" if bad else "OK")
