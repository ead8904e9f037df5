#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports tulip-control/polytope read-only from /root/reference (scipy/HiGHS backend --
cvxopt/GLPK is not installed in this image), calls the reference's own functions on seeded
inputs and stores inputs + outputs as small .npz files.  Fixtures are data only; no
reference source is copied.

Sets:
  g1_lp.npz        raw lpsolve() triples for the LP forms F1/F2/F3 + generic       (solvers.py:76-106)
  g2_reduce.npz    reduce(): kept-row masks, reduced (A,b), Chebyshev radius        (polytope.py:1053-1163)
  g3_edge.npz      edge cases of SURVEY A.4 through cheby_ball / bounding_box / lpsolve
  g4_contains.npz  Polytope.contains / Region.contains incl. boundary points        (polytope.py:206-218,732-746)
  g5_setops.npz    region_diff / Region.intersect pieces and is_adjacent matrices on box grids
                   (polytope.py:2117-2282, :815-830, :1827-1866)
  g6_quickhull.npz Facet normals/offsets, distance(), first-facet assignment,
                   get_furthest, and end-to-end hull facet sets                     (quickhull.py)
  g7_known.npz     the known-answer data of the reference's own tests (tests/polytope_test.py)
  g9_overlap.npz   the pair test of Partition.are_disjoint (is_fulldim(region.intersect(other))) and
                   MetricPartition-style adjacency on cell sets with overlaps  (prop2partition.py:123-192,:244-306)
  g10_bbox.npz     bounding_box() of random polytopes that do not contain the origin, incl. unbounded and
                   empty ones                                                       (polytope.py:1314-1411)
  g11_convex.npz   envelope / is_convex / union(check_convex=True) / mldivide / is_adjacent / intersect on random
                   overlapping, touching and separated polytope pairs (d = 2, 3) and on splits of one polytope
                   by a hyperplane (convex unions)            (polytope.py:1414-1464, 988-1014, 1166-1238, 1470-1505)
  g21_convex_more.npz  g11's record for 32 more pairs, d = 4 included (boxes against polytopes, splits)
  g12_config4.npz  BASELINE config 4: region_diff / Region.intersect / adjacency on the 81-cell 3x3x3x3 grid and
                   region_diff + an adjacency sample on the full 1000-cell 10x10x5x2 grid  (polytope.py:2117-2282)
  g13_volume_subset.npz  seeded volume(), is_subset, == / <= / >= on polytopes and Regions (polytope.py:1529-1594, :1032-1050)
  g14_wide_reduce.npz  reduce() / Polytope.intersect() on inputs of more than 64 rows    (polytope.py:1053-1163, :255-275)
  g16_partition.npz  Partition.is_cover / are_disjoint / refines / preserves, MetricPartition.compute_adj on box grids,
                   triangles, multi-member regions, a non-cover and overlapping sets  (prop2partition.py:46-306)
  g17_structured.npz  reduce() keep masks on structured (16,3) polytopes: ties, duplicates, tangent rows, corner cuts (:1053-1163)
  g18_distance_order.npz  quickhull distance() at d = 7..16, bitwise: numpy's sum keeps 8 partial sums from 8 elements on (quickhull.py:117-121)
  g20_reduce_rows32.npz reduce() on 17..32 rows at d = 1..3, incl. the stacks of Polytope.intersect (round 5: reduce_lane_kernel with 32 row slots)
  g19_hull_highdim.npz  quickhull() end to end at d = 8, 9, 12: rows in the reference's order (quickhull.py:141-359)
  g8_hull.npz      quickhull() rows in the reference's own ORDER for seeded RNG, degenerate inputs,
                   qhull() and extreme() vertex sets                                (quickhull.py:141-359,
                   polytope.py:1597-1695)
"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import polytope as pc  # noqa: E402  (the reference)
import polytope.polytope as alg  # noqa: E402
from polytope import solvers  # noqa: E402
from polytope import quickhull as qh  # noqa: E402
from scipy.optimize import linprog  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
assert solvers.default_solver == "scipy", solvers.default_solver


def rand_hpoly(rng, m, d, bounded=True):
    """Rows tangent to spheres of radius 1..2 (origin strictly inside)."""
    A = rng.standard_normal((m, d))
    A /= np.linalg.norm(A, axis=1)[:, None]
    b = 1.0 + rng.random(m)
    if bounded and m >= 2 * d:
        A[:2 * d] = np.vstack([np.eye(d), -np.eye(d)])
        b[:2 * d] = 3.0
    return A, b


def pad(rows, width):
    out = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


# ----------------------------------------------------------------------------- G1
def gen_g1():
    rng = np.random.default_rng(20260928)
    recs = []
    for (m, d) in [(10, 2), (16, 3), (32, 6), (64, 16), (5, 3), (7, 1)]:
        for t in range(48):
            A, b = rand_hpoly(rng, m, d, bounded=(t % 4 != 0))
            G1 = np.c_[A, np.sqrt(np.sum(A * A, axis=1))]
            c1 = -np.r_[np.zeros(d), 1.0]
            k = int(rng.integers(m))
            h2 = b.copy()
            h2[k] += 0.1
            i = int(rng.integers(d))
            e = np.zeros(d)
            e[i] = rng.choice([-1.0, 1.0])
            x0 = rng.standard_normal(d) * 2
            h4 = A @ x0 + rng.random(m) * rng.choice([1.0, 1.0, 1.0, -0.05])
            for form, (c, G, h) in enumerate(
                    [(c1, G1, b), (-A[k], A, h2), (e, A, b), (rng.standard_normal(d), A, h4)]):
                sol = solvers.lpsolve(c, G, h)
                s_np = linprog(c, G, h, None, None, bounds=(None, None), options={"presolve": False})
                recs.append(dict(form=form + 1, m=m, n=G.shape[1], c=c, G=G, h=h,
                                 status=sol["status"], status_nopresolve=s_np.status,
                                 fun=np.nan if sol["fun"] is None else sol["fun"],
                                 x=np.full(G.shape[1], np.nan) if sol["x"] is None else sol["x"]))
    N = len(recs)
    out = dict(
        form=np.array([r["form"] for r in recs], np.int32),
        m=np.array([r["m"] for r in recs], np.int32),
        n=np.array([r["n"] for r in recs], np.int32),
        status=np.array([r["status"] for r in recs], np.int32),
        status_nopresolve=np.array([r["status_nopresolve"] for r in recs], np.int32),
        fun=np.array([r["fun"] for r in recs]),
        c=pad([r["c"] for r in recs], 17),
        x=pad([r["x"] for r in recs], 17),
        h=pad([r["h"] for r in recs], 64),
        G=pad([r["G"].ravel() for r in recs], 64 * 17),
    )
    np.savez_compressed(os.path.join(HERE, "g1_lp.npz"), **out)
    print("g1:", N, "LPs; status histogram", np.bincount(out["status"]))


# ----------------------------------------------------------------------------- G2
def match_rows(Ain, bin_, Aout, bout):
    """Indices of the input rows the reference kept (see module docstring of the tests:
    exact duplicates resolve to the LAST index, as the dedupe at polytope.py:1104-1109 does)."""
    nin = np.sqrt(np.sum(Ain * Ain, axis=1))
    An = Ain / nin[:, None]
    bn = bin_ / nin
    used = []
    for a, bb in zip(Aout, bout):
        err = np.abs(An - a).sum(axis=1) + np.abs(bn - bb)
        cand = np.nonzero(err < 1e-9)[0]
        cand = [c for c in cand if c not in used]
        assert len(cand) >= 1, (a, bb, err.min())
        used.append(cand[-1])
    return sorted(used)


def structured_polys(rng):
    out = []
    # the reference's own test_reduce data (tests/polytope_test.py:601-622)
    out.append((np.array([[1.0, 0.1], [1.0, 0.1], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]]),
                np.array([50.0, 50.5, -40.0, 1.0, 0.0])))
    # boxes with duplicated / scaled-duplicate rows
    for d in (2, 3, 4):
        I = np.vstack([np.eye(d), -np.eye(d)])
        b = np.r_[np.ones(d), np.zeros(d)]
        out.append((np.vstack([I, I]), np.r_[b, b]))
        out.append((np.vstack([I, 2 * I]), np.r_[b, 2 * b + 0.5]))
        out.append((np.vstack([I, I[::-1]]), np.r_[b, b[::-1] + 1e-9]))
    # stacked overlapping / touching squares (Polytope.intersect stacks rows, polytope.py:270-273)
    sq = np.vstack([np.eye(2), -np.eye(2)])
    for off in (0.0, 0.25, 0.5, 1.0 - 1e-6):
        out.append((np.vstack([sq, sq]), np.r_[1, 1, 0, 0, 1 + off, 1 + off, -off, -off].astype(float)))
    # box cut by random redundant and non-redundant planes
    for d in (2, 3, 5):
        for _ in range(4):
            I = np.vstack([np.eye(d), -np.eye(d)])
            R = rng.standard_normal((3 * d, d))
            R /= np.linalg.norm(R, axis=1)[:, None]
            bb = np.r_[np.ones(2 * d), rng.uniform(0.3, 2.5, 3 * d)]
            out.append((np.vstack([I, R]), bb))
    # simplex-like polytopes with m <= d+1 and m = d+2
    for d in (2, 3, 4):
        A = np.vstack([-np.eye(d), np.ones((1, d))])
        out.append((A, np.r_[np.zeros(d), 1.0]))
        out.append((np.vstack([A, np.ones((1, d))]), np.r_[np.zeros(d), 1.0, 2.0]))
    # empty / flat polytopes
    out.append((np.vstack([sq]), np.array([1.0, 1.0, -2.0, 0.0])))
    out.append((np.vstack([sq]), np.array([1.0, 1.0, -1.0, 0.0])))
    return out


def gen_g2():
    rng = np.random.default_rng(77)
    polys = []
    for (m, d, cnt) in [(16, 3, 64), (10, 2, 48), (24, 4, 32), (32, 6, 24), (64, 16, 6), (40, 3, 16)]:
        for t in range(cnt):
            polys.append(rand_hpoly(rng, m, d, bounded=(t % 8 != 7)))
    polys += structured_polys(rng)
    recs = []
    for (A, b) in polys:
        p = pc.Polytope(A.copy(), b.copy())  # normalises (polytope.py:128-138)
        An, bn = p.A.copy(), p.b.copy()
        q = pc.reduce(p)
        m = An.shape[0]
        if q.A.size == 0:
            kept = []
        else:
            kept = match_rows(An, bn, q.A, q.b)
        mask = np.zeros(64, bool)
        mask[kept] = True
        recs.append(dict(m=m, d=An.shape[1], A=An, b=bn, mask=mask, empty=(q.A.size == 0),
                         minrep=bool(q.minrep), r=float(p._chebR),
                         Aout=q.A, bout=q.b))
    out = dict(
        m=np.array([r["m"] for r in recs], np.int32),
        d=np.array([r["d"] for r in recs], np.int32),
        A=pad([r["A"].ravel() for r in recs], 64 * 16),
        b=pad([r["b"] for r in recs], 64),
        mask=np.array([r["mask"] for r in recs]),
        empty=np.array([r["empty"] for r in recs]),
        minrep=np.array([r["minrep"] for r in recs]),
        r=np.array([r["r"] for r in recs]),
        Aout=pad([r["Aout"].ravel() for r in recs], 64 * 16),
        bout=pad([r["bout"] for r in recs], 64),
    )
    np.savez_compressed(os.path.join(HERE, "g2_reduce.npz"), **out)
    print("g2:", len(recs), "polytopes; empty", int(out["empty"].sum()), "minrep", int(out["minrep"].sum()),
          "mean kept", out["mask"].sum(1).mean())


# ----------------------------------------------------------------------------- G3
def gen_g3():
    cases = {
        "halfspace": (np.array([[1.0, 0.0]]), np.array([1.0])),
        "cone": (np.array([[1.0, 0.0], [0.0, 1.0]]), np.array([1.0, 1.0])),
        "slab": (np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([1.0, 1.0])),
        "infeasible_box": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([1.0, -2.0, 1.0, 1.0])),
        "rectangle": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([2.0, 0.0, 1.0, 0.0])),
        "unit_square": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([1.0, 0.0, 1.0, 0.0])),
        "flat": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([1.0, -1.0, 1.0, 0.0])),
        "triangle": (np.array([[-1.0, 0], [0, -1.0], [1.0, 1.0]]), np.array([0.0, 0.0, 1.0])),
        "interval_1d": (np.array([[1.0], [-1.0]]), np.array([1.0, 0.0])),
        "shifted_big": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([1e3 + 1, -1e3, 5e2 + 1, -5e2])),
        "thin_slab_2e-7": (np.array([[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]), np.array([1 + 1e-7, -1 + 1e-7, 1.0, 0.0])),
    }
    out = {}
    names = []
    for name, (A, b) in cases.items():
        p = pc.Polytope(A.copy(), b.copy())
        r, xc = pc.cheby_ball(p)
        G = np.c_[p.A, np.sqrt(np.sum(p.A * p.A, axis=1))]
        sol = solvers.lpsolve(-np.r_[np.zeros(p.dim), 1.0], G, p.b)
        p2 = pc.Polytope(A.copy(), b.copy())
        lb, ub = pc.bounding_box(p2)
        names.append(name)
        out[name + "_A"] = p.A
        out[name + "_b"] = p.b
        out[name + "_r"] = np.float64(r)
        out[name + "_xc"] = np.full(p.dim, np.nan) if xc is None else np.asarray(xc, float)
        out[name + "_f1status"] = np.int32(sol["status"])
        out[name + "_lb"] = lb.ravel()
        out[name + "_ub"] = ub.ravel()
        out[name + "_fulldim"] = np.bool_(pc.is_fulldim(pc.Polytope(A.copy(), b.copy())))
    out["names"] = np.array(names)
    # empty polytope object
    r, xc = pc.cheby_ball(pc.Polytope())
    out["emptyobj_r"] = np.float64(r)
    np.savez_compressed(os.path.join(HERE, "g3_edge.npz"), **out)
    print("g3:", len(names), "edge cases")


# ----------------------------------------------------------------------------- G4
def gen_g4():
    rng = np.random.default_rng(4)
    P, d, m = 16, 3, 12
    A = np.zeros((P, m, d))
    b = np.zeros((P, m))
    for p in range(P):
        Ap, bp = rand_hpoly(rng, m, d)
        cen = rng.uniform(-1, 1, d)
        bp = bp * 0.5 + Ap @ cen
        pp = pc.Polytope(Ap, bp)
        A[p], b[p] = pp.A, pp.b
    N = 4096
    X = rng.uniform(-2.5, 2.5, (d, N))
    # boundary points: project some points onto facets exactly-ish
    for q in range(0, 512):
        p = q % P
        i = q % m
        x = X[:, q]
        x = x - (A[p, i] @ x - b[p, i]) * A[p, i]
        X[:, q] = x
    tols = np.array([0.0, 1e-7, 0.01, 1.2])
    res = np.zeros((len(tols), P, N), bool)
    reg = np.zeros((len(tols), N), bool)
    polys = [pc.Polytope(A[p], b[p], normalize=False) for p in range(P)]
    region = pc.Region(polys)
    for ti, tol in enumerate(tols):
        for p in range(P):
            res[ti, p] = polys[p].contains(X, abs_tol=tol)
        reg[ti] = region.contains(X, abs_tol=tol)
    # axis-aligned box with exact boundary points (region_contains_test, polytope_test.py:261-277)
    box = pc.box2poly([[0.0, 1.0], [0.0, 2.0]])
    Xb = np.array([[-1.0, 0.0, 0.5, 1.0, 2.0, 0.0, 1.0, 0.5], [1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 2.0, 2.0 + 1e-7]])
    boxres = np.array([box.contains(Xb, abs_tol=t) for t in tols])
    np.savez_compressed(os.path.join(HERE, "g4_contains.npz"), A=A, b=b, X=X, tols=tols, res=res, reg=reg,
                        boxA=box.A, boxb=box.b, Xb=Xb, boxres=boxres)
    print("g4: contains", res.shape, "inside fraction", res.mean())



# ----------------------------------------------------------------------------- G5
def grid_cells(shape, lo=0.0, hi=1.0):
    import itertools
    d = len(shape)
    cells = []
    for idx in itertools.product(*[range(n) for n in shape]):
        iv = [[lo + (hi - lo) * idx[k] / shape[k], lo + (hi - lo) * (idx[k] + 1) / shape[k]] for k in range(d)]
        cells.append(pc.box2poly(iv))
    return cells


def pieces_of(x):
    if isinstance(x, pc.Region):
        return list(x.list_poly)
    return [] if x.A.size == 0 else [x]


def gen_g5():
    rng = np.random.default_rng(55)
    out = {}
    cases = []
    for name, shape in [("g2x2", (2, 2)), ("g3x3", (3, 3)), ("g2x2x2", (2, 2, 2)), ("g2x2x2x2", (2, 2, 2, 2)),
                        ("g3x2x2x1", (3, 2, 2, 1))]:
        d = len(shape)
        cells = grid_cells(shape)
        # minuend: bounded random polytope scaled to radius ~0.3 around the grid centre
        A, b = rand_hpoly(rng, 4 * d, d, bounded=False)
        P = pc.Polytope(A, 0.3 * b + A @ (0.5 * np.ones(d)))
        sub = pc.Region(cells[: max(1, len(cells) // 2)])  # subtract half of the cells
        D = alg.region_diff(P.copy(), sub)
        I = pc.Region(cells).intersect(P.copy())
        adj = np.zeros((len(cells), len(cells)), np.int8)
        for i, a in enumerate(cells):
            adj[i, i] = 1
            for j, bb in enumerate(cells[:i]):
                adj[i, j] = adj[j, i] = pc.is_adjacent(a, bb)
        out[name + "_cellsA"] = np.array([c.A for c in cells])
        out[name + "_cellsb"] = np.array([c.b for c in cells])
        out[name + "_PA"], out[name + "_Pb"] = P.A, P.b
        out[name + "_nsub"] = np.int32(len(sub))
        for tag, X in (("diff", D), ("isect", I)):
            ps = pieces_of(X)
            out[f"{name}_{tag}_n"] = np.int32(len(ps))
            out[f"{name}_{tag}_r"] = np.array([float(pc.cheby_ball(q)[0]) for q in ps])
            out[f"{name}_{tag}_m"] = np.array([q.A.shape[0] for q in ps], np.int32)
            out[f"{name}_{tag}_A"] = pad([q.A.ravel() for q in ps], 64 * d) if ps else np.zeros((0, 64 * d))
            out[f"{name}_{tag}_b"] = pad([q.b for q in ps], 64) if ps else np.zeros((0, 64))
        out[name + "_adj"] = adj
        cases.append(name)
        print("g5", name, "diff pieces", len(pieces_of(D)), "isect pieces", len(pieces_of(I)), "adj pairs", int(adj.sum()))
    out["names"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "g5_setops.npz"), **out)

# ----------------------------------------------------------------------------- G6
def gen_g6():
    rng = np.random.default_rng(6)
    out = {}
    for d in (2, 3, 5, 8):
        # a reference-style start simplex: d+1 points, centred, one Facet per omitted vertex (quickhull.py:188-199)
        S = rng.standard_normal((d + 1, d))
        xc = S.mean(axis=0)
        S0 = S - xc
        normals, offsets = [], []
        facets = []
        for i in range(d + 1):
            ind = np.setdiff1d(np.arange(d + 1), [i])
            f = qh.Facet(S0[ind, :])
            facets.append(f)
            normals.append(np.asarray(f.normal).ravel())
            offsets.append(float(np.asarray(f.distance).ravel()[0]))
        normals, offsets = np.array(normals), np.array(offsets)
        N = 2048
        X = rng.standard_normal((N, d)) * 1.5 - xc
        dist_all = np.array([[float(qh.distance(X[q], f)) for f in facets] for q in range(N)])
        # first-facet assignment (quickhull.py:224-245) and get_furthest (:87-102)
        fop = np.full(N, -1, np.int32)
        dist = np.zeros(N)
        for q in range(N):
            for fi in range(d + 1):
                if dist_all[q, fi] > 1e-7:
                    fop[q] = fi
                    dist[q] = dist_all[q, fi]
                    break
        argmax = np.full(d + 1, -1, np.int64)
        for fi, f in enumerate(facets):
            f.outside = [qh.Outside_point(X[q], dist[q]) for q in range(N) if fop[q] == fi]
            idx = [q for q in range(N) if fop[q] == fi]
            if idx:
                pfar = f.get_furthest()
                # identify by coordinates
                hit = [q for q in idx if np.array_equal(X[q], pfar.coordinates)]
                argmax[fi] = hit[0]
        out[f"d{d}_simplex"] = S0
        out[f"d{d}_normals"] = normals
        out[f"d{d}_offsets"] = offsets
        out[f"d{d}_X"] = X
        out[f"d{d}_distall"] = dist_all
        out[f"d{d}_fop"] = fop
        out[f"d{d}_dist"] = dist
        out[f"d{d}_argmax"] = argmax
    # end-to-end hulls (facet sets canonicalised by sorting rounded [A|b] rows)
    for (d, n) in [(2, 200), (3, 300), (4, 120)]:
        P = rng.random((n, d))
        np.random.seed(123)
        A, b, V = qh.quickhull(P)
        Ab = np.c_[A, b]
        Ab = Ab[np.lexsort(np.round(Ab, 9).T[::-1])]
        out[f"hull{d}_P"] = P
        out[f"hull{d}_Ab"] = Ab
        out[f"hull{d}_V"] = V
    np.savez_compressed(os.path.join(HERE, "g6_quickhull.npz"), **out)
    print("g6: quickhull vectors for d=2,3,5,8 and hulls d=2,3,4")


# ----------------------------------------------------------------------------- G7
def gen_g7():
    out = {}
    # test_lpsolve / test_lpsolve_solver_selection_scipy (polytope_test.py:510-548)
    r = solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]), solver="scipy")
    out["lp1d_x"] = r["x"]
    r = solvers.lpsolve(np.array([1.0, 1.0]), np.array([[-1.0, 0], [0, -1.0]]), np.array([1.0, 1.0]))
    out["lp2d_x"] = r["x"]
    # test_reduce (polytope_test.py:601-622)
    a = np.array([[1.0, 0.1], [1.0, 0.1], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
    b = np.array([50.0, 50.5, -40.0, 1.0, 0.0])
    p2 = pc.reduce(pc.Polytope(a, b))
    l, u = p2.bounding_box
    out["reduce_a"], out["reduce_b"], out["reduce_l"], out["reduce_u"] = a, b, l, u
    out["reduce_Aout"], out["reduce_bout"] = p2.A, p2.b
    # operations_test fixtures (polytope_test.py:57-88): unit squares; fulldim / intersect
    A = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]])
    bb = np.array([1.0, 1.0, 0.0, 0.0])
    Ab2 = np.array([[-1.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 1.0], [0.0, -1.0, 0.0]])
    p1 = pc.Polytope(A, bb)
    p2 = pc.Polytope(Ab2[:, 0:2], Ab2[:, 2])
    p3 = p1.intersect(p2)
    p4 = pc.Polytope(np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]]), np.array([0.5, 0.5, 0.5, 0.5]))
    p5 = p2.intersect(p4)
    out["sq_A"], out["sq_b"], out["sq_Ab2"] = A, bb, Ab2
    out["sq_fulldim"] = np.array([pc.is_fulldim(p1), pc.is_fulldim(p2), pc.is_fulldim(pc.Polytope()),
                                  pc.is_fulldim(pc.Polytope(A, bb - 1e3)), pc.is_fulldim(p3),
                                  pc.is_fulldim(p4), pc.is_fulldim(p5)])
    out["sq_p5_A"], out["sq_p5_b"] = p5.A, p5.b
    out["sq_cheb"] = np.r_[p1.chebR, p1.chebXc, p2.chebR, p2.chebXc, p4.chebR, p4.chebXc]
    # is_inside_test (polytope_test.py:279-296)
    box = pc.Polytope.from_box([[0.0, 1.0], [0.0, 2.0]])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["inside"] = np.array([
            pc.is_inside(box, np.array([0.0, 1.0])), pc.is_inside(box, np.array([0.0, 1.0]), 0.01),
            pc.is_inside(box, np.array([2.0, 0.0])), pc.is_inside(box, np.array([2.0, 0.0]), 0.01),
            pc.is_inside(box, np.array([2.0, 0.0]), 1.2)])
    # test_bounding_box_to_polytope boxes (polytope_test.py:299-312)
    for i, iv in enumerate([[[0, 1]], [[0, 1], [0, 2]], [[-1, 2], [3, 5], [-5, -3]]]):
        p = pc.box2poly(iv)
        l, u = p.bounding_box
        out[f"bbox{i}_A"], out[f"bbox{i}_b"], out[f"bbox{i}_l"], out[f"bbox{i}_u"] = p.A, p.b, l, u
    np.savez_compressed(os.path.join(HERE, "g7_known.npz"), **out)
    print("g7: known-answer data")


# ----------------------------------------------------------------------------- G8
def gen_g8():
    rng = np.random.default_rng(8)
    out = {}
    # end-to-end hulls, rows kept in the order the reference returns them (global RNG seeded)
    cases = [(2, 150, 11), (3, 400, 12), (4, 150, 13), (5, 60, 14), (3, 4, 15), (2, 3, 16)]
    for k, (d, n, seed) in enumerate(cases):
        P = rng.standard_normal((n, d)) if k % 2 else rng.random((n, d))
        np.random.seed(seed)
        A, b, V = qh.quickhull(P)
        out[f"hull{k}_P"], out[f"hull{k}_seed"] = P, np.array(seed)
        out[f"hull{k}_A"], out[f"hull{k}_b"], out[f"hull{k}_V"] = A, b, V
    out["hull_ncases"] = np.array(len(cases))
    # degenerate: unit cube corners + interior points (coplanar points: facets are triangulated);
    # compared as the set of distinct hyperplanes
    corners = np.array([[i, j, k] for i in (0.0, 1.0) for j in (0.0, 1.0) for k in (0.0, 1.0)])
    P = np.vstack([corners, 0.2 + 0.6 * rng.random((40, 3))])
    np.random.seed(21)
    A, b, V = qh.quickhull(P)
    out["cube_P"], out["cube_A"], out["cube_b"], out["cube_V"] = P, A, b, V
    # too few points / flat input -> empty
    A, b, V = qh.quickhull(rng.random((3, 3)))
    out["few_Asize"] = np.array(A.size)
    flat = np.c_[rng.random((20, 2)), np.zeros(20)]
    A, b, V = qh.quickhull(flat)
    out["flat_P"], out["flat_Asize"] = flat, np.array(A.size)
    # extreme(): d = 1, 2 (angle sort), 3, 4 (dual hull) on seeded bounded polytopes
    ext = []
    for (d, m, seed) in [(1, 2, 0), (2, 7, 1), (2, 12, 2), (3, 10, 3), (3, 16, 4), (4, 14, 5)]:
        r2 = np.random.default_rng(100 + seed)
        if d == 1:
            Ap, bp = np.array([[1.0], [-1.0]]), np.array([2.0, 1.0])
        else:
            G = r2.standard_normal((m - 2 * d, d))
            G /= np.linalg.norm(G, axis=1)[:, None]
            Ap = np.vstack([np.eye(d), -np.eye(d), G])
            bp = np.r_[np.full(2 * d, 3.0), 1.0 + r2.random(m - 2 * d)]
        np.random.seed(50 + seed)
        poly = pc.Polytope(Ap, bp)
        V = pc.extreme(poly)
        V = V[np.lexsort(np.round(V, 9).T[::-1])]
        k = len(ext)
        out[f"ext{k}_A"], out[f"ext{k}_b"], out[f"ext{k}_V"] = Ap, bp, V
        ext.append(k)
    out["ext_ncases"] = np.array(len(ext))
    # qhull(): Polytope of a point cloud; the known square of the reference's tests
    sq = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0], [0.5, 0.5]])
    np.random.seed(3)
    q = pc.qhull(sq)
    out["qsq_P"], out["qsq_A"], out["qsq_b"], out["qsq_V"] = sq, q.A, q.b, q.vertices
    # BASELINE configs[0]: examples/randplot.py with N = 10 -- sample points in the unit square, qhull,
    # extreme; plus reduce + cheby_ball of the hull (SURVEY 8d, C1 "plumbing")
    np.random.seed(10)
    Vr = np.random.rand(10, 2)
    Pr = pc.qhull(Vr)
    out["randplot_V"], out["randplot_A"], out["randplot_b"] = Vr, Pr.A, Pr.b
    out["randplot_extreme"] = pc.extreme(Pr)
    Pr2 = pc.reduce(pc.Polytope(Pr.A, Pr.b))
    out["randplot_reduced_Ab"] = np.c_[Pr2.A, Pr2.b]
    rr, xx = pc.cheby_ball(Pr2)
    out["randplot_cheb"] = np.r_[rr, xx]
    np.savez_compressed(os.path.join(HERE, "g8_hull.npz"), **out)
    print("g8: ordered hulls, degenerate cube, extreme() d=1..4, qhull square")


# ----------------------------------------------------------------------------- G9
def gen_g9():
    import itertools
    rng = np.random.default_rng(9)
    out = {}
    sets = {}
    # disjoint grid, 2-D and 3-D
    sets["grid2"] = [pc.box2poly([[i, i + 1], [j, j + 1]]) for i in range(4) for j in range(3)]
    sets["grid3"] = [pc.box2poly([[i, i + 1], [j, j + 1], [k, k + 1]]) for i, j, k in itertools.product(range(3), range(2), range(2))]
    # boxes at random offsets: some overlap, some touch, some are apart
    lo = np.round(rng.random((14, 2)) * 3, 1)
    sets["rand2"] = [pc.box2poly([[a, a + 1.0], [c, c + 0.7]]) for a, c in lo]
    lo3 = np.round(rng.random((12, 3)) * 2, 1)
    sets["rand3"] = [pc.box2poly([[a, a + 0.9], [c, c + 0.9], [e, e + 0.9]]) for a, c, e in lo3]
    for name, cells in sets.items():
        n = len(cells)
        over = np.eye(n, dtype=bool)
        adj = np.eye(n, dtype=np.int8)
        for i in range(n):
            for j in range(i):
                f = bool(pc.is_fulldim(cells[i].intersect(cells[j])))     # prop2partition.py:149
                over[i, j] = over[j, i] = f
                adj[i, j] = adj[j, i] = pc.is_adjacent(cells[i], cells[j])  # :266
        out[name + "_A"] = np.array([c.A for c in cells])
        out[name + "_b"] = np.array([c.b for c in cells])
        out[name + "_over"] = over
        out[name + "_adj"] = adj
        print("g9", name, "overlapping pairs", int(over.sum() - n) // 2, "adjacent pairs", int(adj.sum() - n) // 2)
    out["names"] = np.array(list(sets))
    # separate(): two touching squares + one apart + one touching the third -> components (polytope.py:1795-1824)
    sq = lambda x, y: pc.box2poly([[x, x + 1.0], [y, y + 1.0]])
    reg = pc.Region([sq(0, 0), sq(3, 0), sq(1, 0), sq(3, 1), sq(6, 6)])
    comps = alg.separate(reg)
    out["sep_boxes"] = np.array([[0, 0], [3, 0], [1, 0], [3, 1], [6, 6]], dtype=float)
    out["sep_sizes"] = np.array([len(c) for c in comps])
    out["sep_first_b"] = np.array([c.list_poly[0].b for c in comps])
    # is_interior(): the reference's own semantics (:1888-1909)
    big, small, edge = pc.box2poly([[0, 4], [0, 4]]), pc.box2poly([[1, 2], [1, 2]]), pc.box2poly([[0, 1], [1, 2]])
    out["interior"] = np.array([alg.is_interior(big, small), alg.is_interior(big, edge), alg.is_interior(small, big),
                                alg.is_interior(pc.Region([big]), pc.Region([small, edge]))])
    # simplices2polytopes(): a 2-triangle mesh of the unit square (:2419-2439)
    pts = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])
    tri = np.array([[0, 1, 2], [0, 2, 3]])
    np.random.seed(4)
    polys = alg.simplices2polytopes(pts, tri)
    out["mesh_pts"], out["mesh_tri"] = pts, tri
    out["mesh_Ab"] = np.array([np.c_[p.A, p.b][np.lexsort(np.round(np.c_[p.A, p.b], 9).T[::-1])] for p in polys])
    np.savez_compressed(os.path.join(HERE, "g9_overlap.npz"), **out)


def gen_g10():
    """bounding_box (polytope.py:1314-1411) on polytopes away from the origin (the generic LPs need a phase 1
    there): bounded, unbounded (-inf / +inf corners) and empty (the status-2 branch: l = 0, u = l)."""
    rng = np.random.default_rng(10)
    recs = []
    for (m, d, cnt) in [(8, 2, 16), (16, 3, 24), (20, 4, 12), (24, 6, 10), (40, 8, 6)]:
        for k in range(cnt):
            A, b = rand_hpoly(rng, m, d, bounded=(k % 5 != 3))
            if k % 5 == 3:
                A, b = A[:d], b[:d]                       # d half-spaces: an unbounded cone
            cen = 3.0 * rng.standard_normal(d)
            b = b + A @ cen
            if k % 7 == 5:
                b[0] -= 8.0                               # empty
            P = pc.Polytope(A.copy(), b.copy(), normalize=False)
            lo, hi = pc.bounding_box(P)
            recs.append((A, b, np.asarray(lo).ravel(), np.asarray(hi).ravel()))
    mmax = max(r[0].shape[0] for r in recs)
    dmax = max(r[0].shape[1] for r in recs)
    out = dict(m=np.array([r[0].shape[0] for r in recs]), d=np.array([r[0].shape[1] for r in recs]),
               A=pad([r[0].ravel() for r in recs], mmax * dmax), b=pad([r[1] for r in recs], mmax),
               lb=pad([r[2] for r in recs], dmax), ub=pad([r[3] for r in recs], dmax))
    np.savez_compressed(os.path.join(HERE, "g10_bbox.npz"), **out)
    n_inf = int(sum(np.isinf(r[2]).any() or np.isinf(r[3]).any() for r in recs))
    n_empty = int(sum((r[2] == 0).all() and (r[3] == 0).all() for r in recs))
    print("g10: %d polytopes, %d with infinite corners, %d empty" % (len(recs), n_inf, n_empty))


def gen_g11():
    rng = np.random.default_rng(11)
    out = {}
    names = []

    def store(tag, P, Q):
        R = pc.Region([P.copy(), Q.copy()])
        env = alg.envelope(pc.Region([P.copy(), Q.copy()]))
        convex = bool(alg.is_convex(pc.Region([P.copy(), Q.copy()]))[0])
        U = alg.union(P.copy(), Q.copy(), check_convex=True)
        D = alg.mldivide(P.copy(), Q.copy())
        I = P.copy().intersect(Q.copy())
        out[tag + "_PA"], out[tag + "_Pb"], out[tag + "_QA"], out[tag + "_Qb"] = P.A, P.b, Q.A, Q.b
        out[tag + "_convex"] = np.int8(convex)
        out[tag + "_adjacent"] = np.int8(bool(pc.is_adjacent(P.copy(), Q.copy())))
        for key, X in (("env", env), ("union", U), ("diff", D), ("isect", I)):
            ps = pieces_of(X)
            out[f"{tag}_{key}_n"] = np.int32(len(ps))
            out[f"{tag}_{key}_m"] = np.array([q.A.shape[0] for q in ps], np.int32)
            out[f"{tag}_{key}_Ab"] = pad([np.c_[q.A, q.b].ravel() for q in ps], 64 * (P.A.shape[1] + 1)) if ps \
                else np.zeros((0, 64 * (P.A.shape[1] + 1)))
            out[f"{tag}_{key}_r"] = np.array([float(pc.cheby_ball(q)[0]) for q in ps])
        names.append(tag)
        print("g11", tag, "convex", convex, "union", len(pieces_of(U)), "diff", len(pieces_of(D)), "env rows",
              0 if env.A.size == 0 else env.A.shape[0])

    k = 0
    for d in (2, 3):
        for trial in range(5):
            A1, b1 = rand_hpoly(rng, 4 * d + 2, d, bounded=True)
            A2, b2 = rand_hpoly(rng, 3 * d + 2, d, bounded=True)
            shift = [0.4, 1.5, 2.9, 4.5, 9.0][trial] * np.eye(d)[0] + 0.2 * rng.standard_normal(d)
            P = pc.Polytope(A1, 0.6 * b1)
            Q = pc.Polytope(A2, 0.5 * b2 + A2 @ shift)
            store("pair%d" % k, P, Q)
            k += 1
        for trial in range(3):   # one polytope cut in two by a hyperplane through its interior: the union is convex
            A, b = rand_hpoly(rng, 4 * d + 1, d, bounded=True)
            n = rng.standard_normal(d)
            n /= np.linalg.norm(n)
            c = 0.3 * rng.standard_normal()
            P = pc.reduce(pc.Polytope(np.vstack([A, n]), np.r_[0.7 * b, c]))
            Q = pc.reduce(pc.Polytope(np.vstack([A, -n]), np.r_[0.7 * b, -c]))
            store("split%d" % k, P, Q)
            k += 1
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "g11_convex.npz"), **out)


def gen_g21():
    """g11's record for more pairs, d = 4 included (g11: 16 pairs at d = 2, 3): overlapping / touching / separated random
    polytopes, boxes against polytopes, hyperplane splits.  Same layout as g11_convex.npz."""
    rng = np.random.default_rng(21)
    out = {}
    names = []

    def store(tag, P, Q):
        env = alg.envelope(pc.Region([P.copy(), Q.copy()]))
        convex = bool(alg.is_convex(pc.Region([P.copy(), Q.copy()]))[0])
        U = alg.union(P.copy(), Q.copy(), check_convex=True)
        D = alg.mldivide(P.copy(), Q.copy())
        I = P.copy().intersect(Q.copy())
        out[tag + "_PA"], out[tag + "_Pb"], out[tag + "_QA"], out[tag + "_Qb"] = P.A, P.b, Q.A, Q.b
        out[tag + "_convex"] = np.int8(convex)
        out[tag + "_adjacent"] = np.int8(bool(pc.is_adjacent(P.copy(), Q.copy())))
        for key, X in (("env", env), ("union", U), ("diff", D), ("isect", I)):
            ps = pieces_of(X)
            out[f"{tag}_{key}_n"] = np.int32(len(ps))
            out[f"{tag}_{key}_m"] = np.array([q.A.shape[0] for q in ps], np.int32)
            out[f"{tag}_{key}_Ab"] = pad([np.c_[q.A, q.b].ravel() for q in ps], 64 * (P.A.shape[1] + 1)) if ps \
                else np.zeros((0, 64 * (P.A.shape[1] + 1)))
            out[f"{tag}_{key}_r"] = np.array([float(pc.cheby_ball(q)[0]) for q in ps])
        names.append(tag)
        print("g21", tag, "d", P.A.shape[1], "convex", convex, "union", len(pieces_of(U)), "diff", len(pieces_of(D)), flush=True)

    k = 0
    for d, npairs, nsplit in ((2, 12, 4), (3, 7, 3), (4, 4, 2)):
        for trial in range(npairs):
            A1, b1 = rand_hpoly(rng, 3 * d + 2 + int(rng.integers(0, 4)), d, bounded=True)
            if trial % 3 == 2:     # a box against a polytope
                A2 = np.vstack([np.eye(d), -np.eye(d)])
                b2 = np.r_[rng.uniform(0.3, 0.9, d), rng.uniform(0.3, 0.9, d)]
            else:
                A2, b2 = rand_hpoly(rng, 2 * d + 2 + int(rng.integers(0, 4)), d, bounded=True)
                b2 = 0.5 * b2
            shift = float(rng.choice([0.0, 0.3, 0.8, 1.4, 2.2, 5.0])) * np.eye(d)[int(rng.integers(0, d))] + 0.15 * rng.standard_normal(d)
            P = pc.Polytope(A1, 0.6 * b1)
            Q = pc.Polytope(A2, b2 + A2 @ shift)
            store("pair%d" % k, P, Q)
            k += 1
        for trial in range(nsplit):
            A, b = rand_hpoly(rng, 3 * d + 1, d, bounded=True)
            n = rng.standard_normal(d)
            n /= np.linalg.norm(n)
            c = 0.25 * rng.standard_normal()
            P = pc.reduce(pc.Polytope(np.vstack([A, n]), np.r_[0.7 * b, c]))
            Q = pc.reduce(pc.Polytope(np.vstack([A, -n]), np.r_[0.7 * b, -c]))
            store("split%d" % k, P, Q)
            k += 1
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "g21_convex_more.npz"), **out)


# ----------------------------------------------------------------------------- G12
def _store_pieces(out, key, ps, d):
    """Pieces of a Region in the reference's order: row counts, [A|b] rows (NaN padded), Chebyshev radii."""
    width = max([q.A.shape[0] for q in ps] + [1])
    out[key + "_n"] = np.int32(len(ps))
    out[key + "_m"] = np.array([q.A.shape[0] for q in ps], np.int32)
    out[key + "_Ab"] = pad([np.c_[q.A, q.b].ravel() for q in ps], width * (d + 1)) if ps else np.zeros((0, d + 1))
    out[key + "_r"] = np.array([float(pc.cheby_ball(q)[0]) for q in ps])


def gen_g12():
    """BASELINE config 4 (region_diff / find_adjacent_regions on box grids in d = 4, polytope.py:2117-2282,
    prop2partition.py:46-63) at the sizes SURVEY 8c asks for: the 3x3x3x3 grid (81 cells) completely, and the
    full 10x10x5x2 grid (1000 cells): region_diff of a polytope against the 500 cells with x0 < 0.5 (the
    reference needs ~150 s and 88 000 LPs, stacks of up to 73 rows) plus is_adjacent on a 3000-pair sample."""
    import time
    rng = np.random.default_rng(1212)
    out = {}
    # ---- 81 cells
    shape = (3, 3, 3, 3)
    d = 4
    cells = grid_cells(shape)
    A, b = rand_hpoly(rng, 16, d, bounded=False)
    P = pc.Polytope(A, 0.3 * b + A @ (0.5 * np.ones(d)))
    sub = pc.Region(cells[:40])
    calls = []
    orig = alg.lpsolve
    alg.lpsolve = lambda c, G, h, solver=None: (calls.append(G.shape[0]), orig(c, G, h, solver))[1]
    t0 = time.time()
    D = alg.region_diff(P.copy(), sub)
    out["g81_diff_nlp"], out["g81_diff_maxrows"] = np.int32(len(calls)), np.int32(max(calls))
    print("g12 81 cells: region_diff %d pieces, %d LPs, max rows %d, %.1f s" % (len(pieces_of(D)), len(calls), max(calls), time.time() - t0))
    alg.lpsolve = orig
    t0 = time.time()
    I = pc.Region(cells).intersect(P.copy())
    print("g12 81 cells: Region.intersect %d pieces, %.1f s" % (len(pieces_of(I)), time.time() - t0))
    adj = np.eye(len(cells), dtype=np.int8)
    for i, a in enumerate(cells):
        for j, bb in enumerate(cells[:i]):
            adj[i, j] = adj[j, i] = pc.is_adjacent(a, bb)
    out["g81_cellsA"] = np.array([c.A for c in cells])
    out["g81_cellsb"] = np.array([c.b for c in cells])
    out["g81_PA"], out["g81_Pb"], out["g81_nsub"] = P.A, P.b, np.int32(40)
    _store_pieces(out, "g81_diff", pieces_of(D), d)
    _store_pieces(out, "g81_isect", pieces_of(I), d)
    out["g81_adj"] = adj
    # the reference's volume of the difference with explicit seeds per piece (volume(piece, nsamples, seed))
    out["g81_diff_vol"] = np.array([alg.volume(q.copy(), nsamples=4000, seed=100 + k) for k, q in enumerate(pieces_of(D))])
    # ---- 256 cells (4x4x4x4), seeds tried until no two intersecting cells have radii closer than 1e-9 (see the
    # note on ties below): the largest grid on which the reference's visiting order is well defined
    shape = (4, 4, 4, 4)
    cells = grid_cells(shape)
    for seed in range(100):
        r2 = np.random.default_rng(5000 + seed)
        A, b = rand_hpoly(r2, 16, d, bounded=False)
        P = pc.Polytope(A, 0.22 * b + A @ (0.5 * np.ones(d) + 0.05 * r2.standard_normal(d)))
        Rc = np.array([alg.cheby_ball(pc.Polytope(np.vstack([P.A, c.A]), np.hstack([P.b, c.b])))[0] for c in cells[:128]])
        srt = -np.sort(-Rc[Rc >= 1e-7])
        if len(srt) > 20 and (-np.diff(srt)).min() > 1e-9:
            break
    else:
        raise RuntimeError("no tie-free instance found")
    calls = []
    alg.lpsolve = lambda c, G, h, solver=None: (calls.append(G.shape[0]), orig(c, G, h, solver))[1]
    t0 = time.time()
    D = alg.region_diff(P.copy(), pc.Region(cells[:128]))
    alg.lpsolve = orig
    print("g12 256 cells (seed %d, %d intersecting, min radius gap %.2e): region_diff %d pieces, %d LPs, max rows %d, %.1f s"
          % (seed, len(srt), (-np.diff(srt)).min(), len(pieces_of(D)), len(calls), max(calls), time.time() - t0))
    out["g256_PA"], out["g256_Pb"], out["g256_nsub"] = P.A, P.b, np.int32(128)
    out["g256_diff_nlp"], out["g256_diff_maxrows"] = np.int32(len(calls)), np.int32(max(calls))
    _store_pieces(out, "g256_diff", pieces_of(D), d)
    # ---- 1000 cells (C4)
    shape = (10, 10, 5, 2)
    cells = grid_cells(shape)
    A, b = rand_hpoly(rng, 16, d, bounded=False)
    P = pc.Polytope(A, 0.3 * b + A @ (0.5 * np.ones(d)))
    calls = []
    alg.lpsolve = lambda c, G, h, solver=None: (calls.append(G.shape[0]), orig(c, G, h, solver))[1]
    t0 = time.time()
    D = alg.region_diff(P.copy(), pc.Region(cells[:500]))
    alg.lpsolve = orig
    print("g12 1000 cells: region_diff %d pieces, %d LPs, max rows %d, %.1f s" % (len(pieces_of(D)), len(calls), max(calls), time.time() - t0))
    out["c4_diff_nlp"], out["c4_diff_maxrows"] = np.int32(len(calls)), np.int32(max(calls))
    # region_diff visits the cells in the order argsort(-Rc) of their stacked Chebyshev radii (polytope.py:2145-2157).
    # On a grid hundreds of cells have mathematically EQUAL radii (every cell whose ball is limited by its own
    # facets: 301 of 444 here), so the reference's order among them -- and with it the decomposition -- is decided
    # by the last-bit rounding of its LP solver.  The order it used is recorded so that the search itself can be
    # pinned at this size; Rc is recomputed exactly as :2148-2152 does.
    Rc = np.zeros(500)
    for i, c in enumerate(cells[:500]):
        Rc[i], _ = alg.cheby_ball(pc.Polytope(np.vstack([P.A, c.A]), np.hstack([P.b, c.b])))
    out["c4_Rc"] = Rc
    out["c4_order"] = np.argsort(-Rc).astype(np.int32)
    srt = -np.sort(-Rc[Rc >= 1e-7])
    out["c4_ties_1e12"] = np.int32(int((-np.diff(srt) < 1e-12).sum()))
    print("g12 1000 cells: %d intersecting cells, %d neighbouring radii closer than 1e-12" % (len(srt), int(out["c4_ties_1e12"])))
    out["c4_shape"] = np.array(shape, np.int32)
    out["c4_PA"], out["c4_Pb"], out["c4_nsub"] = P.A, P.b, np.int32(500)
    _store_pieces(out, "c4_diff", pieces_of(D), d)
    # pair sample: every pair among the first 40 cells + 2220 random pairs
    n = len(cells)
    pairs = [(i, j) for i in range(40) for j in range(i)]
    while len(pairs) < 3000:
        i, j = (int(v) for v in rng.integers(0, n, 2))
        if i != j:
            pairs.append((max(i, j), min(i, j)))
    out["c4_pairs"] = np.array(pairs, np.int32)
    out["c4_pairs_adj"] = np.array([pc.is_adjacent(cells[i], cells[j]) for i, j in pairs], np.int8)
    print("g12 1000 cells: %d sampled pairs, %d adjacent" % (len(pairs), int(out["c4_pairs_adj"].sum())))
    np.savez_compressed(os.path.join(HERE, "g12_config4.npz"), **out)


# ----------------------------------------------------------------------------- G13
def gen_g13():
    """volume (polytope.py:1529-1594: seeded via numpy.random.default_rng(seed)), is_subset (:1032-1050),
    == / <= / >= (:220-230, :748-758) on polytopes and Regions."""
    rng = np.random.default_rng(1313)
    out = {}
    vols = []
    k = 0
    for d in (1, 2, 3, 4, 5):
        for trial in range(3):
            A, b = rand_hpoly(rng, max(2 * d, 3 * d + trial), d, bounded=True)
            shift = 0.5 * rng.standard_normal(d)
            P = pc.Polytope(A, (0.4 + 0.3 * trial) * b + A @ shift)
            out["v%d_A" % k], out["v%d_b" % k] = P.A, P.b
            ns = [None, 777, 20000][trial]
            seed = 5 + k
            v = alg.volume(P.copy(), nsamples=ns, seed=seed)
            l, u = P.copy().bounding_box
            # number of samples and of hits, so that a mismatch can be told from a rounding difference of the box
            N = {1: 50, 2: 500, 3: 3000}.get(d, 10000) if ns is None else ns
            out["v%d_nsamples" % k] = np.int64(-1 if ns is None else ns)
            out["v%d_seed" % k] = np.int64(seed)
            out["v%d_vol" % k] = np.float64(v)
            out["v%d_hits" % k] = np.int64(round(v / np.prod(u - l) * N))
            out["v%d_lb" % k], out["v%d_ub" % k] = l.ravel(), u.ravel()
            vols.append(v)
            k += 1
    out["nvol"] = np.int32(k)
    print("g13: %d seeded volumes" % k, np.round(vols, 4))
    # subset / equality tests
    rel = []

    def store(tag, X, Y):
        def pack(Z, key):
            ps = pieces_of(Z)
            dd = ps[0].A.shape[1]
            out[f"{tag}_{key}_n"] = np.int32(len(ps))
            out[f"{tag}_{key}_isreg"] = np.int8(isinstance(Z, pc.Region))
            out[f"{tag}_{key}_m"] = np.array([q.A.shape[0] for q in ps], np.int32)
            out[f"{tag}_{key}_Ab"] = pad([np.c_[q.A, q.b].ravel() for q in ps], max(q.A.shape[0] for q in ps) * (dd + 1))
        pack(X, "X")
        pack(Y, "Y")
        res = [bool(pc.is_subset(X.copy(), Y.copy())), bool(pc.is_subset(Y.copy(), X.copy())),
               bool(X.copy() == Y.copy()), bool(X.copy() <= Y.copy()), bool(X.copy() >= Y.copy()),
               bool(X.copy() != Y.copy())]
        out[tag + "_res"] = np.array(res, np.int8)
        rel.append(tag)
        print("g13", tag, res)

    t = 0
    for d in (2, 3):
        for trial in range(4):
            A1, b1 = rand_hpoly(rng, 4 * d, d, bounded=True)
            P = pc.Polytope(A1, 0.5 * b1)
            if trial == 0:      # shrunk copy: strict subset
                Q = pc.Polytope(A1, 0.4 * b1)
            elif trial == 1:    # the same set with redundant rows added: equal
                A2, b2 = rand_hpoly(rng, 3, d, bounded=False)
                Q = pc.Polytope(np.vstack([A1, A2]), np.r_[0.5 * b1, 5.0 * b2])
            elif trial == 2:    # overlapping, neither contains the other
                Q = pc.Polytope(A1, 0.5 * b1 + A1 @ (0.8 * np.eye(d)[0]))
            else:               # disjoint
                Q = pc.Polytope(A1, 0.5 * b1 + A1 @ (9.0 * np.eye(d)[0]))
            store("rel%d" % t, Q, P)
            t += 1
    # Regions: a box against its own grid cells, a sub-grid, a grid with one cell missing, and polytope vs Region
    for shape in ((2, 2), (3, 2), (2, 2, 2)):
        d = len(shape)
        cells = grid_cells(shape)
        box = pc.box2poly([[0.0, 1.0]] * d)
        store("rel%d" % t, box, pc.Region(cells)); t += 1
        store("rel%d" % t, pc.Region(cells[:-1]), pc.Region(cells)); t += 1
        store("rel%d" % t, pc.Region(cells[:-1]), box); t += 1
        store("rel%d" % t, pc.Region(cells[: len(cells) // 2]), pc.Region(cells[len(cells) // 2:])); t += 1
    out["rel_names"] = np.array(rel)
    np.savez_compressed(os.path.join(HERE, "g13_volume_subset.npz"), **out)


def gen_g14():
    """reduce() and Polytope.intersect() beyond 64 rows (polytope.py:1053-1163 has no row limit; intersect stacks
    m1 + m2 rows, :268-275): kept-row masks of the stacked input, the reduced rows, radii."""
    rng = np.random.default_rng(1414)
    out = {}
    k = 0
    for (m1, m2, d) in [(40, 40, 3), (50, 30, 4), (70, 0, 2), (64, 64, 5), (33, 48, 3), (100, 28, 6), (45, 45, 8)]:
        A1, b1 = rand_hpoly(rng, m1, d, bounded=True)
        P = pc.Polytope(A1, b1)
        if m2:
            A2, b2 = rand_hpoly(rng, m2, d, bounded=True)
            shift = 0.25 * rng.standard_normal(d)
            Q = pc.Polytope(A2, 0.9 * b2 + A2 @ shift)
            R = P.copy().intersect(Q.copy())
            Ain, bin_ = np.vstack([P.A, Q.A]), np.hstack([P.b, Q.b])
            out["c%d_QA" % k], out["c%d_Qb" % k] = Q.A, Q.b
        else:
            R = alg.reduce(P.copy())
            Ain, bin_ = P.A, P.b
        assert Ain.shape[0] > 64
        out["c%d_PA" % k], out["c%d_Pb" % k] = P.A, P.b
        out["c%d_has_Q" % k] = np.int8(1 if m2 else 0)
        out["c%d_A" % k], out["c%d_b" % k] = R.A, R.b
        out["c%d_mask" % k] = match_rows(Ain, bin_, R.A, R.b)
        out["c%d_minrep" % k] = np.int8(bool(R.minrep))
        out["c%d_r" % k] = np.float64(alg.cheby_ball(R)[0])
        print("g14 case %d: %d + %d rows, d = %d -> %d rows, r = %.6f" % (k, m1, m2, d, R.A.shape[0], out["c%d_r" % k]))
        k += 1
    out["n"] = np.int32(k)
    np.savez_compressed(os.path.join(HERE, "g14_wide_reduce.npz"), **out)


def gen_g15():
    """reduce() of the reference on the mid shapes that run one polytope per wavefront in the HIP build (more than 32 rows,
    d = 5..13; polytope.py:1053-1163): same record layout as g2.  Every eighth polytope is half-open."""
    rng = np.random.default_rng(1515)
    recs = []
    for (m, d, cnt) in [(40, 6, 8), (64, 8, 8), (33, 5, 8), (64, 5, 6), (48, 10, 6), (57, 13, 4), (64, 12, 4), (36, 7, 6)]:
        for t in range(cnt):
            A, b = rand_hpoly(rng, m, d, bounded=(t % 8 != 7))
            if t % 4 == 1:   # a duplicated and a slightly shifted row (the dedupe step, :1094-1110)
                A[1], b[1] = A[0], b[0]
                A[3], b[3] = A[2], b[2] + 0.05
            p = pc.Polytope(A.copy(), b.copy())
            An, bn = p.A.copy(), p.b.copy()
            q = pc.reduce(p)
            kept = [] if q.A.size == 0 else match_rows(An, bn, q.A, q.b)
            mask = np.zeros(64, bool)
            mask[kept] = True
            recs.append(dict(m=m, d=d, A=An, b=bn, mask=mask, empty=(q.A.size == 0), minrep=bool(q.minrep),
                             r=float(p._chebR), Aout=q.A, bout=q.b))
    out = dict(
        m=np.array([r["m"] for r in recs], np.int32),
        d=np.array([r["d"] for r in recs], np.int32),
        A=pad([r["A"].ravel() for r in recs], 64 * 16),
        b=pad([r["b"] for r in recs], 64),
        mask=np.array([r["mask"] for r in recs]),
        empty=np.array([r["empty"] for r in recs]),
        minrep=np.array([r["minrep"] for r in recs]),
        r=np.array([r["r"] for r in recs]),
        Aout=pad([r["Aout"].ravel() for r in recs], 64 * 16),
        bout=pad([r["bout"] for r in recs], 64),
    )
    np.savez_compressed(os.path.join(HERE, "g15_reduce_mid.npz"), **out)
    print("g15:", len(recs), "polytopes; empty", int(out["empty"].sum()), "minrep", int(out["minrep"].sum()),
          "mean kept", out["mask"].sum(1).mean())


# ----------------------------------------------------------------------------- G16
def _g16_sets():
    """Cell sets for the Partition / MetricPartition classes: name -> (domain, [region, ...]); a region is a list of
    member polytopes.  Boxes, triangles, multi-member regions, a set that does not cover, sets that overlap."""
    import itertools
    box = pc.box2poly
    sets = {}
    g2 = [[box([[i, i + 1], [j, j + 1]])] for i in range(4) for j in range(3)]
    sets["grid2"] = ([box([[0, 4], [0, 3]])], g2)
    sets["grid2_hole"] = ([box([[0, 4], [0, 3]])], g2[:5] + g2[6:11])               # two cells missing: no cover
    sets["grid2_coarse"] = ([box([[0, 4], [0, 3]])],                                   # L-shaped / multi-member regions
                            [[g2[0][0], g2[1][0], g2[3][0]], [g2[2][0], g2[5][0], g2[4][0]],
                             [box([[2, 4], [0, 3]])]])
    sets["grid2_cols"] = ([box([[0, 4], [0, 3]])], [[box([[i, i + 1], [0, 3]])] for i in range(4)])
    g3 = [[box([[i, i + 1], [j, j + 1], [k, k + 1]])] for i, j, k in itertools.product(range(3), range(2), range(2))]
    sets["grid3"] = ([box([[0, 3], [0, 2], [0, 2]])], g3)
    sets["grid3_slabs"] = ([box([[0, 3], [0, 2], [0, 2]])], [[box([[i, i + 1], [0, 2], [0, 2]])] for i in range(3)])
    rng = np.random.default_rng(16)
    lo = np.round(rng.random((10, 2)) * 3, 1)
    sets["rand2"] = ([box([[0, 4], [0, 4]])], [[box([[a, a + 1.2], [c, c + 0.9]])] for a, c in lo])   # overlaps, no cover
    # triangles: every unit square of a 3 x 2 grid cut along its diagonal (non-box rows, m = 3)
    tri = []
    for i in range(3):
        for j in range(2):
            p00, p10, p11, p01 = [i, j], [i + 1, j], [i + 1, j + 1], [i, j + 1]
            tri.append([alg.qhull(np.array([p00, p10, p11], dtype=float))])
            tri.append([alg.qhull(np.array([p00, p11, p01], dtype=float))])
    sets["tri2"] = ([box([[0, 3], [0, 2]])], tri)
    sets["tri2_squares"] = ([box([[0, 3], [0, 2]])], [[box([[i, i + 1], [j, j + 1]])] for i in range(3) for j in range(2)])
    # covers a two-member domain, with overlap between the covering regions
    sets["overlap_cover"] = ([box([[0, 2], [0, 1]]), box([[2, 3], [0, 1]])],
                             [[box([[0, 1.5], [0, 1]])], [box([[1, 3], [0, 1]])]])
    # 4-D slabs of the unit cube (d = 4, the dimension of BASELINE config 4)
    sets["slab4"] = ([box([[0, 1]] * 4)], [[box([[k / 5, (k + 1) / 5]] + [[0, 1]] * 3)] for k in range(5)])
    return sets


def _g16_partition(cls, domain, regions):
    from polytope.prop2partition import Partition, MetricPartition
    dom = domain[0] if len(domain) == 1 else pc.Region(domain)
    part = dict(Partition=Partition, MetricPartition=MetricPartition)[cls](dom)
    part.domain = dom
    part.regions = [pc.Region(list(r)) for r in regions]
    part.adj = None
    return part


def gen_g16():
    """Class-level outputs of polytope/prop2partition.py: Partition.is_cover / are_disjoint / is_partition / refines /
    preserves (:68-228) and MetricPartition.compute_adj (:231-306), find_adjacent_regions (:46-63)."""
    import logging
    logging.disable(logging.CRITICAL)
    sets = _g16_sets()
    out = {"names": np.array(list(sets))}
    for name, (domain, regions) in sets.items():
        d = domain[0].A.shape[1]
        members = [p for r in regions for p in r]
        owner = np.array([k for k, r in enumerate(regions) for _ in r], np.int32)
        mmax = max(p.A.shape[0] for p in members + domain)
        out[name + "_d"] = np.int32(d)
        out[name + "_owner"] = owner
        out[name + "_m"] = np.array([p.A.shape[0] for p in members], np.int32)
        out[name + "_Ab"] = pad([np.c_[p.A, p.b].ravel() for p in members], mmax * (d + 1))
        out[name + "_dom_m"] = np.array([p.A.shape[0] for p in domain], np.int32)
        out[name + "_dom_Ab"] = pad([np.c_[p.A, p.b].ravel() for p in domain], mmax * (d + 1))
        part = _g16_partition("MetricPartition", domain, regions)
        raised = False
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cover = bool(part.is_cover())
        except AttributeError as e:
            # :110 `logger.Error(msg)`: the branch of a set that is not covered
            assert "Error" in str(e), e
            cover, raised = False, True
        disjoint = bool(part.are_disjoint())
        assert disjoint == bool(part.are_disjoint(check_all=True))
        first_ok = bool(part.compute_adj())
        adj = part.adj.toarray()
        assert bool(part.compute_adj())                     # the matrix agrees with itself
        n = len(regions)
        wrong = part.adj.copy()
        wrong[0, n - 1] = 0.0 if wrong[0, n - 1] else 1.0
        part.adj = wrong
        wrong_ok = bool(part.compute_adj())
        far = pc.find_adjacent_regions(part).toarray()
        out[name + "_cover"] = np.bool_(cover)
        out[name + "_cover_raised"] = np.bool_(raised)
        out[name + "_disjoint"] = np.bool_(disjoint)
        out[name + "_adj"] = adj
        out[name + "_adj_ok"] = np.array([first_ok, wrong_ok])
        out[name + "_far"] = far
        print("g16", name, "n", n, "cover", cover, "(raised)" if raised else "", "disjoint", disjoint,
              "adjacent pairs", int((adj != 0).sum() - n) // 2, "compute_adj", first_ok, wrong_ok)
    # refines (:194-207): every ordered pair of sets over the same domain
    groups = [["grid2", "grid2_hole", "grid2_coarse", "grid2_cols"], ["grid3", "grid3_slabs"], ["tri2", "tri2_squares"]]
    pairs, verdict = [], []
    for grp in groups:
        for a in grp:
            for b in grp:
                pa = _g16_partition("Partition", *sets[a])
                pb = _g16_partition("Partition", *sets[b])
                pairs.append(a + ">" + b)
                verdict.append(bool(pa.refines(pb)))
    out["refines_pairs"] = np.array(pairs)
    out["refines"] = np.array(verdict)
    print("g16 refines:", dict(zip(pairs, verdict)))
    # preserves (:209-228).  The reference builds set(other) of Regions, which its Region does not allow (__eq__
    # without __hash__: TypeError); the expected values are the loop of :219-227 over the reference's own `<=` and
    # `intersect`, with the difference taken by identity.
    def preserves(elements, other):
        for item in elements:
            for superset in item.supersets:
                if not item <= superset:
                    return False
            for other_set in [o for o in other if not any(o is s_ for s_ in item.supersets)]:
                if item.intersect(other_set):
                    return False
        return True
    try:
        set([pc.Region([pc.box2poly([[0, 1], [0, 1]])])])
        out["preserves_ref_hashable"] = np.bool_(True)
    except TypeError:
        out["preserves_ref_hashable"] = np.bool_(False)
    pres = []
    for fine, coarse, shift in [("grid2", "grid2_cols", 0), ("grid2", "grid2_cols", 1), ("tri2", "tri2_squares", 0),
                                ("grid2_coarse", "grid2_cols", 0)]:
        pf = _g16_partition("Partition", *sets[fine])
        pcoarse = _g16_partition("Partition", *sets[coarse])
        nc = len(pcoarse.regions)
        for item in pf.regions:
            # annotate with the coarse element holding the item's Chebyshev centre (shifted by `shift` for a wrong one)
            xc = item.list_poly[0].chebXc
            k = [t for t, big in enumerate(pcoarse.regions) if xc in big][0]
            item.supersets = [pcoarse.regions[(k + shift) % nc]]
        pres.append((fine + ">" + coarse + "+" + str(shift), preserves(pf.regions, pcoarse.regions)))
    out["preserves_cases"] = np.array([c for c, _ in pres])
    out["preserves"] = np.array([v for _, v in pres])
    print("g16 preserves:", pres, "reference Regions hashable:", bool(out["preserves_ref_hashable"]))
    # __str__ of Polytope / Region (polytope.py:150-176, :711-721; pinned by the reference's own test_polytope_str)
    strs = [str(pc.Polytope(np.array([[1]]), np.array([1])))]
    for box in ([[0, 1]], [[0, 1], [0, 2]], [[0, 1], [0, 2], [0, 3]]):
        strs.append(str(pc.box2poly(box)))
    strs.append(str(pc.Region([pc.box2poly([[0, 1], [0, 2]]), pc.box2poly([[1, 2], [0, 2]])])))
    strs.append(str(pc.Polytope(np.array([[0.6, -0.8], [-1.0, 0.0], [0.0, 1.0]]), np.array([1.25, 0.5, 3.0]))))
    out["str_cases"] = np.array(strs)
    logging.disable(logging.NOTSET)
    np.savez_compressed(os.path.join(HERE, "g16_partition.npz"), **out)


# ----------------------------------------------------------------------------- G17
def gen_g17():
    """reduce() (polytope.py:1053-1163) on STRUCTURED (16, 3) polytopes (tests/structured_cases.py: duplicated and
    shifted-parallel rows, vertex fans, tangent rows, slacks within a few abs_tol of the threshold, mutually redundant
    corner cuts, lattice normals, one-ulp twins): which input rows the reference keeps."""
    sys.path.insert(0, os.path.dirname(HERE))
    from structured_cases import structured_polytopes, family_names
    A, b, fam = structured_polytopes(960, seed=17)
    def run(tight):
        """reduce() over the set -> (keep, empty, minrep, r); tight: HiGHS with feasibility tolerances of 1e-10."""
        keep = np.zeros((len(A), 16), bool)
        empty = np.zeros(len(A), bool)
        minrep = np.zeros(len(A), bool)
        rad = np.zeros(len(A))
        saved = alg.lpsolve
        if tight:
            def lp_tight(c, G, h, solver=None):
                sol = linprog(c, G, np.transpose(h), None, None, bounds=(None, None),
                              options={"primal_feasibility_tolerance": 1e-10, "dual_feasibility_tolerance": 1e-10})
                return dict(status=sol.status, x=sol.x, fun=sol.fun)
            alg.lpsolve = lp_tight
        try:
            for k in range(len(A)):
                p = pc.Polytope(A[k], b[k], normalize=False)
                q = pc.reduce(p)
                empty[k] = q.A.size == 0
                minrep[k] = bool(q.minrep)
                rad[k] = float(p._chebR) if p._chebR is not None else np.nan
                if empty[k]:
                    continue
                # Rows come back renormalised by the constructor (A * (1 / norm), :130-138), b after the in-place 0.1
                # round trip (:1149-1151).  Input rows on one plane (twins one ulp apart, exact duplicates) cannot be
                # told apart by value: of such a class the dedupe (:1097-1109) leaves the row of smallest normalised b,
                # the LAST of equals.
                scale = 1 / np.sqrt(np.sum(A[k] * A[k], axis=1))
                An, bn = A[k] * scale[:, None], b[k] * scale
                used = []
                for a, bb in zip(q.A, q.b):
                    cand = np.nonzero((An == a).all(1) & (np.abs(bn - bb) < 1e-12))[0]
                    assert cand.size, (k, a, bb)
                    best = [c for c in cand if bn[c] == bn[cand].min()][-1]
                    assert best not in used
                    used.append(best)
                keep[k, used] = True
        finally:
            alg.lpsolve = saved
        return keep, empty, minrep, rad

    keep, empty, minrep, rad = run(False)
    keep_t, empty_t, minrep_t, rad_t = run(True)
    # HiGHS stops at points that violate rows by up to 1e-7 (its default primal feasibility tolerance), which moves an F2
    # objective by about as much: a row whose exact slack lies that close to abs_tol is kept or dropped by the tolerance
    # setting, not by the polytope.  `pinned`: the reference's answer is the same with the tolerances at 1e-10.
    pinned = (keep == keep_t).all(1) & (empty == empty_t) & (minrep == minrep_t)
    print("g17: verdicts that depend on HiGHS's feasibility tolerance:", int((~pinned).sum()), "of", len(A),
          "by family", np.bincount(fam[~pinned], minlength=len(family_names())))
    np.savez_compressed(os.path.join(HERE, "g17_structured.npz"), seed=np.int64(17), keep=keep, empty=empty, minrep=minrep,
                        r=rad, r_tight=rad_t, pinned=pinned, fam=fam.astype(np.int32), names=np.array(family_names()),
                        A_sum=np.float64(A.sum()), b_sum=np.float64(b.sum()))
    print("g17:", len(A), "structured polytopes; empty", int(empty.sum()), "mean kept", keep.sum(1).mean())


# ----------------------------------------------------------------------------- G18
def gen_g18():
    """distance() (quickhull.py:117-121) where numpy's sum changes its order of additions: below 8 elements np.sum
    adds in index order, from 8 on it keeps eight partial sums (pairwise_sum).  Distances of 4096 points to the d + 1
    facets of a start simplex at d = 7, 8, 9, 12, 16, to be matched BIT FOR BIT; first-facet assignment (:224-245) and
    furthest point (:87-102) with them."""
    rng = np.random.default_rng(18)
    out = {}
    for d in (7, 8, 9, 12, 16):
        S = rng.standard_normal((d + 1, d))
        xc = S.mean(axis=0)
        S0 = S - xc
        facets = [qh.Facet(S0[np.setdiff1d(np.arange(d + 1), [i]), :]) for i in range(d + 1)]
        normals = np.array([np.asarray(f.normal).ravel() for f in facets])
        offsets = np.array([float(np.asarray(f.distance).ravel()[0]) for f in facets])
        N = 4096
        X = rng.standard_normal((N, d)) * 1.5 - xc
        dist_all = np.array([[float(qh.distance(X[q], f)) for f in facets] for q in range(N)])
        fop = np.full(N, -1, np.int32)
        dist = np.zeros(N)
        for q in range(N):
            hit = np.nonzero(dist_all[q] > 1e-7)[0]
            if hit.size:
                fop[q], dist[q] = hit[0], dist_all[q, hit[0]]
        argmax = np.full(d + 1, -1, np.int64)
        for fi, f in enumerate(facets):
            idx = [q for q in range(N) if fop[q] == fi]
            f.outside = [qh.Outside_point(X[q], dist[q]) for q in idx]
            if idx:
                pfar = f.get_furthest()
                argmax[fi] = [q for q in idx if np.array_equal(X[q], pfar.coordinates)][0]
        out.update({f"d{d}_normals": normals, f"d{d}_offsets": offsets, f"d{d}_X": X, f"d{d}_distall": dist_all,
                    f"d{d}_fop": fop, f"d{d}_dist": dist, f"d{d}_argmax": argmax})
    np.savez_compressed(os.path.join(HERE, "g18_distance_order.npz"), **out)
    print("g18: distances at d = 7, 8, 9, 12, 16 (numpy's pairwise order from 8 elements on)")


# ----------------------------------------------------------------------------- G19
def gen_g19():
    """quickhull() end to end in dimensions 8, 9 and 12 -- where distance()'s np.sum adds in eight partial sums (g18) --
    rows in the order the reference returns them under a seeded global RNG (quickhull.py:141-359)."""
    rng = np.random.default_rng(19)
    out = {}
    cases = [(8, 14, 5), (8, 16, 6), (9, 15, 7), (12, 16, 8)]
    for k, (d, n, seed) in enumerate(cases):
        P = rng.standard_normal((n, d))
        np.random.seed(seed)
        A, b, V = qh.quickhull(P)
        out[f"hull{k}_P"], out[f"hull{k}_seed"] = P, np.array(seed)
        out[f"hull{k}_A"], out[f"hull{k}_b"], out[f"hull{k}_V"] = A, b, V
        print("g19 hull", k, "d", d, "points", n, "facets", A.shape[0])
    out["hull_ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "g19_hull_highdim.npz"), **out)


# ----------------------------------------------------------------------------- G20
def gen_g20():
    """reduce() of the reference on polytopes of 17..32 rows in d = 1..3 -- the shapes reduce_lane_kernel takes with 32 row
    slots per polytope (round 5) -- among them the stacks Polytope.intersect builds from two 16-row polytopes
    (polytope.py:268-275: vstack, then reduce): same record layout as g2 / g15."""
    rng = np.random.default_rng(2020)
    recs = []

    def record(A, b, m, d):
        p = pc.Polytope(A.copy(), b.copy())
        An, bn = p.A.copy(), p.b.copy()
        q = pc.reduce(p)
        kept = [] if q.A.size == 0 else match_rows(An, bn, q.A, q.b)
        mask = np.zeros(64, bool)
        mask[kept] = True
        recs.append(dict(m=m, d=d, A=An, b=bn, mask=mask, empty=(q.A.size == 0), minrep=bool(q.minrep),
                         r=float(p._chebR), Aout=q.A, bout=q.b))

    for (m, d, cnt) in [(32, 3, 12), (24, 3, 10), (17, 3, 8), (32, 2, 10), (20, 2, 8), (24, 1, 4)]:
        for t in range(cnt):
            A, b = rand_hpoly(rng, m, d, bounded=(t % 6 != 5))
            if t % 4 == 1:   # a duplicated and a slightly shifted row (the dedupe step, :1094-1110)
                A[1], b[1] = A[0], b[0]
                A[3], b[3] = A[2], b[2] + 0.05
            record(A, b, m, d)
    for t in range(14):       # the stack of P.intersect(Q): two (16,3) polytopes, the second one moved a little
        A1, b1 = rand_hpoly(rng, 16, 3)
        A2, b2 = rand_hpoly(rng, 16, 3)
        shift = 0.4 * rng.standard_normal(3) if t % 5 else 4.0 * np.ones(3)    # (every fifth pair does not meet: empty)
        record(np.vstack([A1, A2]), np.hstack([b1, b2 + A2 @ shift]), 32, 3)
    out = dict(
        m=np.array([r["m"] for r in recs], np.int32),
        d=np.array([r["d"] for r in recs], np.int32),
        A=pad([r["A"].ravel() for r in recs], 64 * 16),
        b=pad([r["b"] for r in recs], 64),
        mask=np.array([r["mask"] for r in recs]),
        empty=np.array([r["empty"] for r in recs]),
        minrep=np.array([r["minrep"] for r in recs]),
        r=np.array([r["r"] for r in recs]),
        Aout=pad([r["Aout"].ravel() for r in recs], 64 * 16),
        bout=pad([r["bout"] for r in recs], 64),
    )
    np.savez_compressed(os.path.join(HERE, "g20_reduce_rows32.npz"), **out)
    print("g20:", len(recs), "polytopes; empty", int(out["empty"].sum()), "minrep", int(out["minrep"].sum()),
          "mean kept", out["mask"].sum(1).mean())


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21"]
    for w in which:
        globals()["gen_" + w]()
