"""The Polytope/Region/solvers mirror (polytope_amd.polytope, polytope_amd.solvers) against the
golden vectors generated from the reference, once per backend:

  backend 'scipy' : CPU; exercises the host orchestration (caches, early-outs, DFS of
                    region_diff, union/envelope) with the reference's own LP arithmetic
  backend 'hip'   : marked gpu; the same assertions with every LP going through the HIP kernels

These read like the reference's own tests (tests/polytope_test.py): same fixtures, same
assertions, plus the numeric pins of tests/golden/g*.npz.
"""
import warnings

import numpy as np
import pytest

from conftest import load_golden

TOL = 1e-9


@pytest.fixture(params=["scipy", pytest.param("hip", marks=pytest.mark.gpu)])
def pc(request):
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    old = solvers.default_solver
    if request.param == "hip":
        assert "hip" in solvers.installed_solvers, "HIP backend not installed on a GPU box"
    solvers.default_solver = request.param
    yield pc
    solvers.default_solver = old


def unpad(A_row, b_row, m, d):
    return A_row[:m * d].reshape(m, d), b_row[:m]


# ------------------------------------------------------------------ solvers (polytope_test.py:510-575)
def test_lpsolve_contract(pc):
    from polytope_amd import solvers
    res = solvers.lpsolve(np.array([1.0, 1.0]), np.array([[-1.0, 0], [0, -1.0]]), np.array([1.0, 1.0]))
    assert res["x"].ndim == 1 and res["x"].shape == (2,)
    res = solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]))
    assert res["x"].shape == (1,) and res["x"] == np.array([-1.0]) and res["status"] == 0
    assert isinstance(res["fun"], float)
    # LP failure is a status, not an exception; x and fun are None
    res = solvers.lpsolve(np.array([1.0]), np.array([[1.0]]), np.array([1.0]))
    assert res["status"] == 3 and res["x"] is None and res["fun"] is None
    with pytest.raises(RuntimeError):
        solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]), solver="glpk")
    with pytest.raises(Exception, match="unknown LP solver"):
        solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]), solver="nope")


def test_hip_missing_raises_not_falls_back():
    """'hip' is opt-in (the default follows the reference's rule, solvers.py:66-73); selected without a GPU it
    must raise -- as a missing GLPK does in the reference (solvers.py:200-207) -- never fall back to a CPU path."""
    from polytope_amd import solvers
    assert solvers.default_solver == "scipy"
    if "hip" in solvers.installed_solvers:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]), solver="hip")
    import polytope_amd.polytope as pcm
    solvers.default_solver = "hip"
    try:
        with pytest.raises(RuntimeError):
            solvers.lpsolve(np.array([1.0]), np.array([[-1.0]]), np.array([1.0]))
        with pytest.raises(RuntimeError):
            pcm.cheby_ball(pcm.box2poly([[0, 1], [0, 1]]))
        with pytest.raises(RuntimeError):
            pcm.box2poly([[0, 1], [0, 1]]).contains(np.zeros((2, 3)))
    finally:
        solvers.default_solver = "scipy"


# ------------------------------------------------------------------ data type (ref :122-148)
def test_polytope_ctor_normalises(pc):
    A = np.array([[3.0, 4.0], [0.0, 0.0], [0.0, -2.0]])
    b = np.array([10.0, 1.0, 4.0])
    p = pc.Polytope(A, b)
    assert p.A.shape == (2, 2)
    assert np.allclose(p.A, [[0.6, 0.8], [0.0, -1.0]]) and np.allclose(p.b, [2.0, 2.0])
    q = pc.Polytope(A, b, normalize=False)
    assert np.array_equal(q.A, A) and np.array_equal(q.b, b)
    assert len(p) == 0 and p.dim == 2 and not p.minrep and p.bbox is None and p.fulldim is None
    e = pc.Polytope()
    assert pc.is_empty(e) and pc.cheby_ball(e) == (0, None) and not pc.is_fulldim(e)
    c = p.copy()
    assert c is not p and np.array_equal(c.A, p.A)
    box = pc.Polytope.from_box([[0.0, 1.0], [0.0, 2.0]])
    assert box.minrep and np.array_equal(box.b, [1.0, 2.0, 0.0, 0.0])
    with pytest.raises(Exception):
        pc.Polytope.from_box([[1.0, 0.0]])
    r = pc.Region([box, pc.Polytope()])
    assert len(r) == 1 and r.dim == 2
    # The reference takes empty members out with list.remove (ref :694-696), i.e. the first element that compares EQUAL to the
    # empty polytope -- and `==` is "both differences have a volume below 1e-7" (ref :220-230, :1032-1050): an earlier member
    # without a volume (the unbounded p; a box of side 1e-3 in R^3) goes instead and the empty one stays.  Mirrored.
    r = pc.Region([p, pc.Polytope()])
    assert len(r) == 1 and pc.is_empty(r.list_poly[0])
    tiny, big = pc.box2poly([[0.0, 1e-3]] * 3), pc.box2poly([[0.0, 1.0]] * 3)
    r = pc.Region([big, tiny, pc.Polytope(), pc.box2poly([[2.0, 3.0]] * 3)])
    assert [q.A.shape[0] for q in r.list_poly] == [6, 0, 6]


# ------------------------------------------------------------------ cheby / bbox edge cases (g3)
def test_cheby_and_bbox_edges(pc):
    g = load_golden("g3_edge.npz")
    for name in g["names"]:
        p = pc.Polytope(g[f"{name}_A"], g[f"{name}_b"], normalize=False)
        r, xc = pc.cheby_ball(p)
        assert abs(r - float(g[f"{name}_r"])) <= TOL, name
        assert (xc is None) == bool(np.isnan(g[f"{name}_xc"]).all()), name
        if xc is not None:
            nrm = np.sqrt((p.A * p.A).sum(1))
            assert np.max(p.A @ xc + nrm * r - p.b) <= 1e-9, name
            assert p._chebR == r and p._chebXc is xc  # cached on success only (ref :1298-1299)
        else:
            assert p._chebXc is None
        l, u = pc.bounding_box(pc.Polytope(g[f"{name}_A"], g[f"{name}_b"], normalize=False))
        assert l.shape == (p.dim, 1) and u.shape == (p.dim, 1)
        assert np.allclose(l.ravel(), g[f"{name}_lb"], atol=TOL, rtol=0), name
        assert np.allclose(u.ravel(), g[f"{name}_ub"], atol=TOL, rtol=0), name
        if abs(r - 1e-7) > 1e-12:
            assert pc.is_fulldim(pc.Polytope(g[f"{name}_A"], g[f"{name}_b"], normalize=False)) == bool(
                g[f"{name}_fulldim"]), name


def test_known_answers(pc):
    g = load_golden("g7_known.npz")
    # test_reduce (polytope_test.py:601-622)
    p2 = pc.reduce(pc.Polytope(g["reduce_a"], g["reduce_b"]))
    l, u = p2.bounding_box
    assert np.allclose(l, [[40.0], [0.0]], rtol=1e-7, atol=1e-7) and np.allclose(u, [[50.0], [1.0]], rtol=1e-7, atol=1e-7)
    assert np.allclose(p2.A, g["reduce_Aout"], atol=1e-12) and np.allclose(p2.b, g["reduce_bout"], atol=1e-12)
    # operations_test squares (polytope_test.py:205-242)
    A, b, Ab2 = g["sq_A"], g["sq_b"], g["sq_Ab2"]
    p1, p2 = pc.Polytope(A, b), pc.Polytope(Ab2[:, 0:2], Ab2[:, 2])
    p3 = p1.intersect(p2)
    p4 = pc.Polytope(np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]]), np.array([0.5, 0.5, 0.5, 0.5]))
    p5 = p2.intersect(p4)
    got = [pc.is_fulldim(p1), pc.is_fulldim(p2), pc.is_fulldim(pc.Polytope()),
           pc.is_fulldim(pc.Polytope(A, b - 1e3)), pc.is_fulldim(p3), pc.is_fulldim(p4), pc.is_fulldim(p5)]
    assert got == list(g["sq_fulldim"])
    assert np.allclose(p5.A, g["sq_p5_A"], atol=1e-12) and np.allclose(p5.b, g["sq_p5_b"], atol=1e-12)
    cheb = np.r_[p1.chebR, p1.chebXc, p2.chebR, p2.chebXc, p4.chebR, p4.chebXc]
    assert np.allclose(cheb, g["sq_cheb"], atol=TOL)  # unit squares: unique centres
    # region_full_dim_test (:211-224)
    reg = pc.Region([p1, p2])
    assert pc.is_fulldim(reg) and not pc.is_fulldim(pc.Region())
    # is_inside_test (:279-296)
    box = pc.Polytope.from_box([[0.0, 1.0], [0.0, 2.0]])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = [pc.is_inside(box, np.array([0.0, 1.0])), pc.is_inside(box, np.array([0.0, 1.0]), 0.01),
               pc.is_inside(box, np.array([2.0, 0.0])), pc.is_inside(box, np.array([2.0, 0.0]), 0.01),
               pc.is_inside(box, np.array([2.0, 0.0]), 1.2)]
        assert got == list(g["inside"])
        region = pc.Region([box])
        assert pc.is_inside(region, np.array([0.0, 1.0])) and not pc.is_inside(region, np.array([2.0, 0.0]))
        assert pc.is_inside(region, np.array([2.0, 0.0]), 1.2)
    # bounding boxes of boxes (:299-312)
    for i in range(3):
        p = pc.Polytope(g[f"bbox{i}_A"], g[f"bbox{i}_b"])
        l, u = p.bounding_box
        assert np.allclose(l, g[f"bbox{i}_l"], atol=TOL) and np.allclose(u, g[f"bbox{i}_u"], atol=TOL)


def test_contains_semantics(pc):
    # polytope_contains_test / region_contains_test (polytope_test.py:244-277)
    g = load_golden("g7_known.npz")
    p = pc.Polytope(g["sq_A"], g["sq_b"])
    assert [0.1, 0.3] in p and [2, 0] not in p
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.random((2, 8)) - np.array([[0], [1]]), rng.random((2, 8))], axis=1)
    assert np.array_equal(p.contains(pts), np.array([False] * 8 + [True] * 8))
    poly = pc.Polytope(np.array([[1.0], [-1.0]]), np.array([1.0, 0.0]))
    reg = pc.Region([poly])
    assert 0.5 in reg
    points = np.array([[-1.0, 0.0, 0.5, 1.0, 2.0]])
    assert np.array_equal(reg.contains(points), [False, True, True, True, False])
    assert np.array_equal(reg.contains(points, abs_tol=0), [False, False, True, False, False])
    with pytest.raises(ValueError):
        reg.contains(np.zeros((2, 3)))
    g4 = load_golden("g4_contains.npz")
    polys = [pc.Polytope(g4["A"][k], g4["b"][k], normalize=False) for k in range(g4["A"].shape[0])]
    region = pc.Region(polys)
    for ti, tol in enumerate(g4["tols"]):
        assert np.array_equal(region.contains(g4["X"], abs_tol=float(tol)), g4["reg"][ti])
        assert np.array_equal(polys[3].contains(g4["X"], abs_tol=float(tol)), g4["res"][ti, 3])


# ------------------------------------------------------------------ reduce (g2, g15)
def test_reduce_golden_mid_shapes(pc):
    """g15: the reference's reduce() on 33..64 rows, d = 5..13 (the shapes the HIP build runs one polytope per wavefront)."""
    g = load_golden("g15_reduce_mid.npz")
    for i in range(0, len(g["m"]), 2):
        m, d = int(g["m"][i]), int(g["d"][i])
        A, b = unpad(g["A"][i], g["b"][i], m, d)
        p = pc.Polytope(A, b, normalize=False)
        q = pc.reduce(p)
        assert (q.A.size == 0) == bool(g["empty"][i]), i
        k = q.A.shape[0]
        Aout, bout = unpad(g["Aout"][i], g["bout"][i], k, d)
        assert k == int(g["mask"][i].sum()), (i, m, d, k, int(g["mask"][i].sum()))
        assert np.allclose(q.A, Aout, atol=1e-12, rtol=0) and np.allclose(q.b, bout, atol=1e-12, rtol=0), i
        assert bool(q.minrep) == bool(g["minrep"][i]) and abs(p.chebR - g["r"][i]) <= TOL, i


def test_reduce_golden(pc):
    g = load_golden("g2_reduce.npz")
    for i in range(len(g["m"])):
        m, d = int(g["m"][i]), int(g["d"][i])
        if m == 64 and i % 2:
            continue  # keep the CPU run short
        A, b = unpad(g["A"][i], g["b"][i], m, d)
        p = pc.Polytope(A, b, normalize=False)
        q = pc.reduce(p)
        assert (q.A.size == 0) == bool(g["empty"][i]), i
        if g["empty"][i]:
            assert p.fulldim is not None and not p.fulldim
            continue
        k = q.A.shape[0]
        Aout, bout = unpad(g["Aout"][i], g["bout"][i], k, d)
        assert k == int(g["mask"][i].sum()), i
        assert np.allclose(q.A, Aout, atol=1e-12, rtol=0) and np.allclose(q.b, bout, atol=1e-12, rtol=0), i
        assert bool(q.minrep) == bool(g["minrep"][i]), i
        assert abs(p.chebR - g["r"][i]) <= TOL, i
        assert pc.reduce(q) is q or not q.minrep
    # Region: member-wise, flat members dropped (ref :1068-1077)
    polys = []
    for i in (0, 1, 2, 70, 71):
        m, d = int(g["m"][i]), int(g["d"][i])
        if d == int(g["d"][0]):
            polys.append(pc.Polytope(*unpad(g["A"][i], g["b"][i], m, d), normalize=False))
    red = pc.reduce(pc.Region(polys))
    assert isinstance(red, pc.Region) and all(p.minrep for p in red)


# ------------------------------------------------------------------ set operations (g5)
def _pieces(pc, x):
    if isinstance(x, pc.Region):
        return list(x.list_poly)
    return [] if x.A.size == 0 else [x]


@pytest.mark.parametrize("name", ["g2x2", "g3x3", "g2x2x2", "g2x2x2x2", "g3x2x2x1"])
def test_region_diff_intersect_adjacency(pc, name):
    g = load_golden("g5_setops.npz")
    cells = [pc.Polytope(A, b, normalize=False) for A, b in zip(g[name + "_cellsA"], g[name + "_cellsb"])]
    for c in cells:
        c.minrep = True  # as box2poly builds them
    d = cells[0].dim
    P = pc.Polytope(g[name + "_PA"], g[name + "_Pb"], normalize=False)
    sub = pc.Region(cells[: int(g[name + "_nsub"])])
    D = pc.region_diff(P.copy(), sub)
    for tag, X in (("diff", D),):
        ps = _pieces(pc, X)
        assert len(ps) == int(g[f"{name}_{tag}_n"]), (name, tag, len(ps))
        for k, q in enumerate(ps):  # same pieces, same order, same rows
            m = int(g[f"{name}_{tag}_m"][k])
            Ak, bk = unpad(g[f"{name}_{tag}_A"][k], g[f"{name}_{tag}_b"][k], m, d)
            assert q.A.shape[0] == m, (name, tag, k)
            assert np.allclose(q.A, Ak, atol=1e-9, rtol=0) and np.allclose(q.b, bk, atol=1e-9, rtol=0), (name, tag, k)
            assert abs(float(pc.cheby_ball(q)[0]) - g[f"{name}_{tag}_r"][k]) <= TOL
    # adjacency of all cell pairs: one batch (find_adjacent_regions, prop2partition.py:46-63)
    n = len(cells)
    pairs = [(cells[i], cells[j]) for i in range(n) for j in range(i)]
    flags = pc.is_adjacent_pairs(pairs)
    adj = np.eye(n, dtype=np.int8)
    for (i, j), f in zip([(i, j) for i in range(n) for j in range(i)], flags):
        adj[i, j] = adj[j, i] = f
    assert np.array_equal(adj, g[name + "_adj"])
    assert pc.is_adjacent(cells[0], cells[1]) == bool(g[name + "_adj"][0, 1])
    assert pc.is_adjacent(pc.Region(cells[:2]), cells[-1]) == bool(
        g[name + "_adj"][0, n - 1] or g[name + "_adj"][1, n - 1])


@pytest.mark.parametrize("name", ["g2x2", "g2x2x2"])
def test_region_intersect_merges_convex(pc, name):
    g = load_golden("g5_setops.npz")
    cells = [pc.Polytope(A, b, normalize=False) for A, b in zip(g[name + "_cellsA"], g[name + "_cellsb"])]
    for c in cells:
        c.minrep = True
    d = cells[0].dim
    P = pc.Polytope(g[name + "_PA"], g[name + "_Pb"], normalize=False)
    I = pc.Region(cells).intersect(P.copy())
    ps = _pieces(pc, I)
    assert len(ps) == int(g[f"{name}_isect_n"])
    rs = sorted(float(pc.cheby_ball(q)[0]) for q in ps)
    assert np.allclose(rs, sorted(g[f"{name}_isect_r"]), atol=1e-7)
    for k, q in enumerate(ps):  # same pieces in the same order as the reference produced
        m = int(g[f"{name}_isect_m"][k])
        Ak, bk = unpad(g[f"{name}_isect_A"][k], g[f"{name}_isect_b"][k], m, d)
        assert q.A.shape[0] == m
        assert np.allclose(q.A, Ak, atol=1e-9, rtol=0) and np.allclose(q.b, bk, atol=1e-9, rtol=0), (name, k)
    # the pieces tile P restricted to the grid [0,1]^d
    rng = np.random.default_rng(1)
    X = rng.random((d, 2000))
    assert np.array_equal(I.contains(X, abs_tol=0), P.contains(X, abs_tol=0) & pc.Region(cells).contains(X, abs_tol=0))


def _g11_names():
    return [("g11_convex.npz", str(n)) for n in load_golden("g11_convex.npz")["names"]] + \
           [("g21_convex_more.npz", str(n)) for n in load_golden("g21_convex_more.npz")["names"]]


@pytest.mark.parametrize("fixture,name", _g11_names())
def test_envelope_convexity_union_vs_reference(pc, fixture, name):
    """g11 / g21: envelope / is_convex / union(check_convex=True) / mldivide / intersect / is_adjacent of polytope pairs as
    the reference computes them (polytope.py:1414-1464, 988-1014, 1166-1238, 1470-1505): overlapping, touching,
    separated pairs, boxes against polytopes and hyperplane splits (convex unions), d = 2, 3 (g11) and 2..4 (g21).  Same pieces
    in the same order, rows within 1e-9."""
    g = load_golden(fixture)
    if fixture.startswith("g21") and pc.solvers.default_solver == "scipy" and name != "pair27" and \
            list(g["names"]).index(name) % 3 != 0:
        pytest.skip("g21 on the scipy backend: every third pair + the list.remove case (the whole fixture runs on 'hip')")
    P = pc.Polytope(g[name + "_PA"], g[name + "_Pb"], normalize=False)
    Q = pc.Polytope(g[name + "_QA"], g[name + "_Qb"], normalize=False)
    d = P.dim
    got = {
        "env": pc.envelope(pc.Region([P.copy(), Q.copy()])),
        "union": pc.union(P.copy(), Q.copy(), check_convex=True),
        "diff": pc.mldivide(P.copy(), Q.copy()),
        "isect": P.copy().intersect(Q.copy()),
    }
    assert bool(pc.is_convex(pc.Region([P.copy(), Q.copy()]))[0]) == bool(g[name + "_convex"])
    assert bool(pc.is_adjacent(P.copy(), Q.copy())) == bool(g[name + "_adjacent"])
    for key, X in got.items():
        ps = _pieces(pc, X)
        assert len(ps) == int(g[f"{name}_{key}_n"]), (name, key, len(ps), int(g[f"{name}_{key}_n"]))
        for k, q in enumerate(ps):
            m = int(g[f"{name}_{key}_m"][k])
            Ab = g[f"{name}_{key}_Ab"][k][:m * (d + 1)].reshape(m, d + 1)
            assert q.A.shape[0] == m, (name, key, k, q.A.shape[0], m)
            assert np.allclose(np.c_[q.A, q.b], Ab, rtol=0, atol=1e-9), (name, key, k)
            assert abs(float(pc.cheby_ball(q)[0]) - g[f"{name}_{key}_r"][k]) <= TOL, (name, key, k)


def test_mldivide_and_subset(pc):
    a = pc.box2poly([[0.0, 2.0], [0.0, 1.0]])
    b = pc.box2poly([[1.0, 3.0], [0.0, 1.0]])
    d = a.diff(b)
    ps = _pieces(pc, d)
    assert len(ps) == 1
    l, u = ps[0].bounding_box
    assert np.allclose(l.ravel(), [0, 0], atol=1e-7) and np.allclose(u.ravel(), [1, 1], atol=1e-7)
    assert pc.is_subset(pc.box2poly([[0.5, 1.0], [0.2, 0.8]]), a)
    assert not pc.is_subset(b, a)
    assert a == pc.box2poly([[0.0, 2.0], [0.0, 1.0]]) and a != b
    # covered polytope -> empty difference
    assert pc.is_empty(pc.box2poly([[0.2, 0.4], [0.2, 0.4]]).diff(a)) or not pc.is_fulldim(
        pc.box2poly([[0.2, 0.4], [0.2, 0.4]]).diff(a))
    # volume: seeded, reproducible; box volume within Monte-Carlo error
    v = pc.volume(pc.box2poly([[0.0, 2.0], [0.0, 1.0]]), nsamples=2000, seed=3)
    assert v == pytest.approx(2.0, rel=0.1)
    with pytest.raises(ValueError):
        pc.volume(a, nsamples=0)


@pytest.mark.parametrize("name", ["g3x3", "g2x2x2x2"])
def test_find_adjacent_regions(pc, name):
    """prop2partition.find_adjacent_regions (prop2partition.py:46-63): all pair LPs in one batch."""
    from polytope_amd import prop2partition as p2p
    g = load_golden("g5_setops.npz")
    cells = [pc.Polytope(A, b, normalize=False) for A, b in zip(g[name + "_cellsA"], g[name + "_cellsb"])]
    adj = p2p.find_adjacent_regions([pc.Region([c]) for c in cells])
    assert adj.shape == (len(cells), len(cells)) and adj.dtype == np.int8
    assert np.array_equal(adj.toarray(), g[name + "_adj"])
    # mixed shapes take the general path
    mixed = [pc.Region([cells[0], cells[1]]), cells[2], pc.Region([cells[3]])]
    adj2 = p2p.find_adjacent_regions(mixed).toarray()
    want01 = int(g[name + "_adj"][0, 2] or g[name + "_adj"][1, 2])
    assert adj2[0, 1] == want01 == adj2[1, 0] and adj2[2, 2] == 1


def test_find_adjacent_regions_high_dimension(pc):
    """The same in 9 dimensions (a 3 x 2 x 2 x 1^5 x 2 grid of boxes; on the HIP backend the pair LPs of d >= 9 run one pair
    per wavefront, adjacent_w_kernel): neighbours in the grid -- faces, edges, corners -- and nothing else; disjoint."""
    import itertools
    from polytope_amd import prop2partition as p2p
    shape = (3, 2, 2, 1, 1, 1, 1, 1, 2)
    lo = np.array(list(itertools.product(*[range(k) for k in shape])), dtype=float)
    cells = [pc.box2poly(np.c_[l, l + 1.0].tolist()) for l in lo]
    adj = p2p.find_adjacent_regions([pc.Region([c]) for c in cells]).toarray()
    want = (np.abs(lo[:, None, :] - lo[None, :, :]).max(axis=2) <= 1).astype(np.int8)
    assert np.array_equal(adj, want)
    assert p2p.are_disjoint(cells)


# ------------------------------------------------------------------ Partition.are_disjoint / compute_adj
@pytest.mark.parametrize("name", ["grid2", "grid3", "rand2", "rand3"])
def test_overlap_and_compute_adj(pc, name):
    """prop2partition.are_disjoint's pair test (is_fulldim(region.intersect(other)), ref :146-149) and
    MetricPartition.compute_adj (:244-306) for all pairs at once, against the reference's own loops."""
    from polytope_amd import prop2partition as p2p
    g = load_golden("g9_overlap.npz")
    cells = [pc.Polytope(A, b) for A, b in zip(g[name + "_A"], g[name + "_b"])]
    over = p2p.overlap_matrix_dense(cells)
    assert np.array_equal(over, g[name + "_over"])
    want_disjoint = not (g[name + "_over"] & ~np.eye(len(cells), dtype=bool)).any()
    assert p2p.are_disjoint(cells) == want_disjoint == p2p.are_disjoint(cells, check_all=True)
    adj, ok = p2p.compute_adj([pc.Region([c]) for c in cells])
    assert ok and np.array_equal(adj.toarray() != 0, g[name + "_adj"] != 0)
    _, ok2 = p2p.compute_adj(cells, previous=adj)
    assert ok2
    wrong = adj.copy()
    wrong[0, len(cells) - 1] = 0 if wrong[0, len(cells) - 1] else 1
    assert not p2p.compute_adj(cells, previous=wrong)[1]
    # regions with several member polytopes: overlap if any member pair overlaps
    regs = [pc.Region(cells[:2]), pc.Region(cells[2:4]), cells[-1]]
    o3 = p2p.overlap_matrix_dense(regs)
    go = g[name + "_over"]
    assert o3[0, 1] == go[:2, 2:4].any() and o3[0, 2] == go[:2, -1].any() and o3[1, 2] == go[2:4, -1].any()


def test_separate_interior_simplices(pc):
    """The remaining small callers of the LP path: separate (ref :1795-1824), is_interior (:1888-1909, with the
    reference's own semantics) and simplices2polytopes (:2419-2439), against the reference's outputs."""
    g = load_golden("g9_overlap.npz")
    sq = lambda x, y: pc.box2poly([[x, x + 1.0], [y, y + 1.0]])  # noqa: E731
    reg = pc.Region([sq(*xy) for xy in g["sep_boxes"]])
    comps = pc.separate(reg)
    assert [len(c) for c in comps] == list(g["sep_sizes"])
    assert np.allclose(np.array([c.list_poly[0].b for c in comps]), g["sep_first_b"])
    big, small, edge = pc.box2poly([[0, 4], [0, 4]]), pc.box2poly([[1, 2], [1, 2]]), pc.box2poly([[0, 1], [1, 2]])
    got = [pc.is_interior(big, small), pc.is_interior(big, edge), pc.is_interior(small, big),
           pc.is_interior(pc.Region([big]), pc.Region([small, edge]))]
    assert got == [bool(v) for v in g["interior"]]
    np.random.seed(4)
    polys = pc.simplices2polytopes(g["mesh_pts"], g["mesh_tri"])
    for p, want in zip(polys, g["mesh_Ab"]):
        Ab = np.c_[p.A, p.b]
        assert np.allclose(Ab[np.lexsort(np.round(Ab, 9).T[::-1])], want, rtol=0, atol=1e-12)


# ------------------------------------------------------------------ BASELINE config 4 (g12) and f3 (g13)
def _pieces_from(g, key, d):
    out = []
    for k in range(int(g[key + "_n"])):
        m = int(g[key + "_m"][k])
        Ab = g[key + "_Ab"][k][:m * (d + 1)].reshape(m, d + 1)
        out.append((Ab[:, :d], Ab[:, d], float(g[key + "_r"][k])))
    return out


def _grid_cells(pc, shape):
    import itertools
    d = len(shape)
    return [pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(d)])
            for idx in itertools.product(*[range(n) for n in shape])]


def _assert_same_pieces(pc, got, want, tag):
    assert len(got) == len(want), (tag, len(got), len(want))
    for k, (q, (Ak, bk, rk)) in enumerate(zip(got, want)):  # same pieces, same order, same rows
        assert q.A.shape == Ak.shape, (tag, k, q.A.shape, Ak.shape)
        assert np.allclose(q.A, Ak, atol=1e-9, rtol=0) and np.allclose(q.b, bk, atol=1e-9, rtol=0), (tag, k)
    radii = pc.cheby_ball(pc.Region([q.copy() for q in got]))  # one batch; fills the members' caches
    del radii
    for k, (q, (_, _, rk)) in enumerate(zip(got, want)):
        assert abs(float(pc.cheby_ball(q)[0]) - rk) <= TOL, (tag, k)


def test_config4_grid81_region_diff_and_adjacency(pc):
    """3x3x3x3 grid (81 cells, d = 4), SURVEY 8c G5: region_diff against 40 cells -- pieces, order, rows and radii as
    the reference returns them (polytope.py:2117-2282) -- and the adjacency matrix of all 3240 cell pairs
    (prop2partition.py:46-63 over polytope.py:1827-1866)."""
    g = load_golden("g12_config4.npz")
    cells = [pc.Polytope(A, b, normalize=False) for A, b in zip(g["g81_cellsA"], g["g81_cellsb"])]
    for c in cells:
        c.minrep = True
    P = pc.Polytope(g["g81_PA"], g["g81_Pb"], normalize=False)
    D = pc.region_diff(P.copy(), pc.Region(cells[: int(g["g81_nsub"])]))
    _assert_same_pieces(pc, _pieces(pc, D), _pieces_from(g, "g81_diff", 4), "g81_diff")
    # seeded volumes of the pieces (polytope.py:1529-1594): same default_rng stream -> same hit counts
    for k, q in enumerate(_pieces(pc, D)):
        v = pc.volume(q.copy(), nsamples=4000, seed=100 + k)
        assert v == pytest.approx(float(g["g81_diff_vol"][k]), rel=1e-9, abs=1e-12), k
    from polytope_amd import prop2partition as p2p
    adj = p2p.find_adjacent_regions([pc.Region([c]) for c in cells]).toarray()
    assert np.array_equal(adj, g["g81_adj"])


def test_config4_grid256_region_diff(pc):
    """4x4x4x4 grid (256 cells), region_diff against 128 cells on an instance without near-equal stacked radii, i.e.
    the largest grid on which the reference's visiting order (argsort(-Rc), polytope.py:2153-2157) does not hinge on
    LP rounding noise: pieces, order, rows and radii through the public call."""
    g = load_golden("g12_config4.npz")
    cells = _grid_cells(pc, (4, 4, 4, 4))
    P = pc.Polytope(g["g256_PA"], g["g256_Pb"], normalize=False)
    D = pc.region_diff(P.copy(), pc.Region(cells[: int(g["g256_nsub"])]))
    _assert_same_pieces(pc, _pieces(pc, D), _pieces_from(g, "g256_diff", 4), "g256_diff")


@pytest.mark.gpu
def test_config4_grid81_region_intersect():
    """Region(81 cells).intersect(P): the reference's greedy union(check_convex=True) merge (polytope.py:815-830,
    :1166-1238) -- same pieces in the same order.  (The scipy backend needs ~90 s for this one: GPU only.)"""
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers
    g = load_golden("g12_config4.npz")
    old, solvers.default_solver = solvers.default_solver, "hip"
    try:
        cells = [pcm.Polytope(A, b, normalize=False) for A, b in zip(g["g81_cellsA"], g["g81_cellsb"])]
        for c in cells:
            c.minrep = True
        P = pcm.Polytope(g["g81_PA"], g["g81_Pb"], normalize=False)
        I = pcm.Region(cells).intersect(P.copy())
        _assert_same_pieces(pcm, _pieces(pcm, I), _pieces_from(g, "g81_isect", 4), "g81_isect")
    finally:
        solvers.default_solver = old


def test_config4_full_size_region_diff_reference_backend():
    """BASELINE configs[3] at its stated size through the PUBLIC call on the reference's own LP backend (scipy: the radii,
    and with them the tie order among the 301 cells of equal radius, are the reference's): the 234 pieces of the fixture,
    same order, rows and radii.  The Python-side search (_DiffSearch) issues ~13 000 LPs for it where the reference
    issues 99 039 (cells an ancestor scan found empty are not solved again): ~30 s on one core."""
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers
    g = load_golden("g12_config4.npz")
    old, solvers.default_solver = solvers.default_solver, "scipy"
    try:
        cells = _grid_cells(pcm, tuple(int(v) for v in g["c4_shape"]))
        P = pcm.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
        D = pcm.region_diff(P.copy(), pcm.Region(cells[: int(g["c4_nsub"])]))
        _assert_same_pieces(pcm, _pieces(pcm, D), _pieces_from(g, "c4_diff", 4), "c4_diff")
    finally:
        solvers.default_solver = old


@pytest.mark.gpu
def test_config4_full_size_region_diff_and_adjacency():
    """BASELINE configs[3] at its stated size: the 10x10x5x2 grid of 1000 box cells in d = 4.

    region_diff(P, the 500 cells with x0 < 0.5): the reference needs 99 039 LPs (stacks of up to 69 rows, beyond the
    64-row register engines) and returns 234 pieces; pieces, order, rows and radii must be the reference's.
    find_adjacent_regions over the 1000 single-cell Regions (499 500 pair LPs of shape (16,5)): equal to the grid's
    Chebyshev-distance-1 neighbourhood, and to the reference's is_adjacent on a 3000-pair sample."""
    import itertools
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers, prop2partition as p2p
    from polytope_amd.batch import overlap_pairs
    g = load_golden("g12_config4.npz")
    shape = tuple(int(v) for v in g["c4_shape"])
    old, solvers.default_solver = solvers.default_solver, "hip"
    try:
        cells = _grid_cells(pcm, shape)
        P = pcm.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
        sub = pcm.Region(cells[: int(g["c4_nsub"])])
        # (i) the search pinned against the reference.  301 of the 444 intersecting cells have mathematically equal
        # stacked radii (balls limited by the cell's own facets); the reference visits them in the order its LP
        # solver's last-bit rounding produced (fixture c4_order, c4_ties_1e12), which no other LP code can
        # reproduce, so that order is handed in; everything else -- 99 039 LPs with stacks of up to 69 rows, the
        # 234 pieces, their order, rows and radii -- must then be the reference's.
        assert int(g["c4_ties_1e12"]) > 100 and int(g["c4_diff_maxrows"]) > 64
        D = pcm.region_diff(P.copy(), sub, _order=g["c4_order"])
        _assert_same_pieces(pcm, _pieces(pcm, D), _pieces_from(g, "c4_diff", 4), "c4_diff")
        # (ii) the public call (own tie order) gives another decomposition of the same kind.  What can be asserted of
        # it is only what the REFERENCE's output satisfies: its 234 pieces lie inside P and are full-dimensional, but
        # they overlap each other (up to 6 deep), reach into subtracted cells and leave parts of P \ sub uncovered
        # (the leaf branch of ref :2225-2245 advances INDICES without advancing `counter`) -- behaviour this build
        # reproduces piece for piece above instead of correcting.
        D2 = pcm.region_diff(P.copy(), sub)
        rng = np.random.default_rng(4)
        l, u = P.bounding_box
        X = l + rng.random((4, 200000)) * (u - l)
        inP = P.contains(X, abs_tol=0)
        for Dk in (D, D2):
            ps = _pieces(pcm, Dk)
            assert len(ps) > 100
            inD = np.zeros(X.shape[1], bool)
            for q in ps:
                inD |= q.contains(X, abs_tol=0)
                assert float(pcm.cheby_ball(q)[0]) > 1e-7
            assert not np.any(inD & ~inP)
        # (iii) as SETS (the public call against the reference's 234 pieces).  They are not the same set and cannot
        # be: neither is P \\ sub -- the reference's decomposition depends on the order of its tied cells, misses 0.105 of the true
        # difference's volume of 1.339 and covers 0.009 outside it (scripts/debug/c4_set_check.py, 2 M seeded points through the
        # containment kernel).  What is asserted: the public call's pieces miss and overshoot the TRUE difference by no more,
        # in total, than the reference's own pieces do (measured: 0.080 against 0.114).
        insub = sub.contains(X, abs_tol=0)
        truth = inP & ~insub
        err = []
        for Dk in (D2, D):        # D reproduces the reference's pieces (i)
            inD = np.zeros(X.shape[1], bool)
            for q in _pieces(pcm, Dk):
                inD |= q.contains(X, abs_tol=0)
            err.append(float(np.mean(truth ^ inD)))
        assert err[0] <= err[1] * 1.05 and err[1] > 0.0, err
        D3 = pcm.region_diff(P.copy(), sub)   # deterministic
        assert [q.A.shape for q in _pieces(pcm, D3)] == [q.A.shape for q in _pieces(pcm, D2)]
        # adjacency
        adj = p2p.find_adjacent_regions([pcm.Region([c]) for c in cells]).toarray()
        index = np.array(list(itertools.product(*[range(n) for n in shape])))
        want = (np.abs(index[:, None, :] - index[None, :, :]).max(axis=2) <= 1).astype(np.int8)
        assert np.array_equal(adj, want)
        pairs = g["c4_pairs"]
        assert np.array_equal(adj[pairs[:, 0], pairs[:, 1]], g["c4_pairs_adj"])
        # Partition.are_disjoint's pair test at this size: no two cells overlap
        A = np.array([c.A for c in cells])
        b = np.array([c.b for c in cells])
        assert np.array_equal(overlap_pairs(A, b), np.eye(len(cells), dtype=np.uint8))
    finally:
        solvers.default_solver = old


def test_volume_seeded_matches_reference(pc):
    """volume(P, nsamples, seed) draws from numpy.random.default_rng(seed) like the reference (polytope.py:1529-1594),
    so the hit counts are the reference's and the value differs only by the rounding of the bounding box."""
    g = load_golden("g13_volume_subset.npz")
    for k in range(int(g["nvol"])):
        P = pc.Polytope(g["v%d_A" % k], g["v%d_b" % k], normalize=False)
        ns = int(g["v%d_nsamples" % k])
        v = pc.volume(P, nsamples=None if ns < 0 else ns, seed=int(g["v%d_seed" % k]))
        l, u = P.bounding_box
        assert np.allclose(l.ravel(), g["v%d_lb" % k], atol=1e-9) and np.allclose(u.ravel(), g["v%d_ub" % k], atol=1e-9)
        d = P.A.shape[1]
        N = ({1: 50, 2: 500, 3: 3000}.get(d, 10000)) if ns < 0 else ns
        assert int(round(v / np.prod(u - l) * N)) == int(g["v%d_hits" % k]), k
        assert v == pytest.approx(float(g["v%d_vol" % k]), rel=1e-9), k
        assert P.volume == v  # cached (ref :382-385)


def test_subset_and_comparisons_match_reference(pc):
    """is_subset (polytope.py:1032-1050) and == / <= / >= / != (:220-230, :748-758) on 20 polytope / Region pairs."""
    g = load_golden("g13_volume_subset.npz")

    def build(tag, key):
        ps = [(int(g[f"{tag}_{key}_m"][k]), g[f"{tag}_{key}_Ab"][k]) for k in range(int(g[f"{tag}_{key}_n"]))]
        mmax = max(m for m, _ in ps)   # the fixture rows hold mmax * (d + 1) numbers
        d = g[f"{tag}_{key}_Ab"].shape[1] // mmax - 1
        polys = []
        for m, row in ps:
            Ab = row[:m * (d + 1)].reshape(m, d + 1)
            polys.append(pc.Polytope(Ab[:, :d], Ab[:, d], normalize=False))
        return pc.Region(polys) if int(g[f"{tag}_{key}_isreg"]) else polys[0]

    for tag in g["rel_names"]:
        tag = str(tag)
        want = [bool(v) for v in g[tag + "_res"]]
        X, Y = build(tag, "X"), build(tag, "Y")
        got = [bool(pc.is_subset(X.copy(), Y.copy())), bool(pc.is_subset(Y.copy(), X.copy())),
               bool(X.copy() == Y.copy()), bool(X.copy() <= Y.copy()), bool(X.copy() >= Y.copy()),
               bool(X.copy() != Y.copy())]
        assert got == want, (tag, got, want)


@pytest.mark.gpu
def test_high_dimensional_box_and_reduce_match_the_scipy_backend():
    """d = 12: bounding_box and reduce of random polytopes with box rows on the 'hip' backend (fused kernels that keep no
    dictionary for the 2d / per-row LPs) against the same calls on the reference's scipy backend -- boxes within 1e-9,
    identical rows kept; one half-open polytope (+inf side) and one empty one."""
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    rng = np.random.default_rng(12)
    d, m = 12, 40
    polys = []
    for k in range(6):
        A = rng.standard_normal((m, d))
        A /= np.linalg.norm(A, axis=1, keepdims=True)
        b = 1.0 + rng.random(m)
        A[:2 * d] = np.vstack([np.eye(d), -np.eye(d)])
        b[:2 * d] = 1.2 + rng.random(2 * d)
        if k == 4:
            A, b = A[1:2 * d].copy(), b[1:2 * d].copy()          # the box without x_0 <= ..: unbounded above
        if k == 5:
            b[30] = -40.0                                        # empty
        polys.append((A, b))
    out = {}
    old = solvers.default_solver
    try:
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            res = []
            for A, b in polys:
                p = pc.Polytope(A.copy(), b.copy())
                l, u = pc.bounding_box(p)
                q = pc.reduce(pc.Polytope(A.copy(), b.copy()))
                res.append((l.ravel(), u.ravel(), q.A.copy(), q.b.copy()))
            out[backend] = res
    finally:
        solvers.default_solver = old
    for (l0, u0, A0, b0), (l1, u1, A1, b1) in zip(out["scipy"], out["hip"]):
        assert np.allclose(l0, l1, rtol=0, atol=TOL) and np.allclose(u0, u1, rtol=0, atol=TOL)
        assert A0.shape == A1.shape and np.allclose(A0, A1, rtol=0, atol=1e-12) and np.allclose(b0, b1, rtol=0, atol=1e-12)
        assert np.array_equal(np.isinf(u0), np.isinf(u1)) and np.array_equal(np.isinf(l0), np.isinf(l1))
    assert np.isinf(out["hip"][4][1][0]) and out["hip"][5][2].size == 0


@pytest.mark.gpu
def test_stacks_beyond_64_rows_through_the_python_layer():
    """intersect / reduce / cheby_ball / bounding_box / is_adjacent on polytopes whose stacks pass 64 rows (the fused
    kernels' limit): the 'hip' backend takes the LDS-resident LP engine for them and must agree with the scipy backend
    (the reference's arithmetic) -- same kept rows, radii and boxes to 1e-9."""
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers
    rng = np.random.default_rng(70)

    def rand_poly(m, d, shift=0.0):
        A = rng.standard_normal((m, d))
        A /= np.linalg.norm(A, axis=1)[:, None]
        return A, 1.0 + 0.3 * rng.random(m) + A @ (shift * np.ones(d))

    cases = [(rand_poly(40, 3), rand_poly(45, 3, 0.2)), (rand_poly(70, 2), rand_poly(10, 2, 0.1)),
             (rand_poly(50, 4), rand_poly(50, 4, -0.15))]
    out = {}
    for backend in ("scipy", "hip"):
        old, solvers.default_solver = solvers.default_solver, backend
        try:
            res = []
            for (A1, b1), (A2, b2) in cases:
                P, Q = pcm.Polytope(A1, b1), pcm.Polytope(A2, b2)
                I = P.intersect(Q)                       # stack of 85 / 80 / 100 rows -> reduce
                big = pcm.Polytope(np.vstack([A1, A2]), np.r_[b1, b2])
                r, xc = pcm.cheby_ball(big)
                l, u = big.bounding_box
                res.append((I.A.copy(), I.b.copy(), float(r), l.ravel().copy(), u.ravel().copy(),
                            bool(pcm.is_adjacent(P, Q)), bool(pcm.is_fulldim(big))))
            out[backend] = res
        finally:
            solvers.default_solver = old
    for (Ia, ib, r, l, u, adj, fd), (Ja, jb, r2, l2, u2, adj2, fd2) in zip(out["scipy"], out["hip"]):
        assert Ia.shape == Ja.shape and np.allclose(Ia, Ja, atol=1e-9, rtol=0) and np.allclose(ib, jb, atol=1e-9, rtol=0)
        assert abs(r - r2) <= TOL and np.allclose(l, l2, atol=1e-9) and np.allclose(u, u2, atol=1e-9)
        assert adj == adj2 and fd == fd2
        assert Ia.shape[0] > 3


def test_region_diff_keeps_a_cell_whose_lp_ended_without_a_verdict(monkeypatch):
    """The search drops a cell from the scans below a node only when its ball LP was SOLVED with a radius <= tol / 2.
    An LP that ends with another status (iteration limit, numerical trouble, unbounded ball) reads as radius 0 in the
    reference (polytope.py:1294-1297), which solves the cell again at every node below: one transient failure must
    not remove the cell from the whole subtree."""
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    monkeypatch.setattr(solvers, "default_solver", "scipy")
    P = pc.box2poly([[0.0, 3.0], [0.0, 1.0]])
    cells = [pc.box2poly([[0.0, 1.0], [0.0, 1.0]]), pc.box2poly([[1.0, 2.0], [0.0, 1.0]])]
    want = pc.region_diff(P.copy(), pc.Region([c.copy() for c in cells]))
    assert isinstance(want, pc.Polytope) or len(want) == 1
    lo, hi = pc.bounding_box(want)
    assert np.allclose(lo.ravel(), [2.0, 0.0]) and np.allclose(hi.ravel(), [3.0, 1.0])
    real, hits = pc.lpsolve, []

    def flaky(c, G, h, solver=None):
        # the root scan's stack for the second cell: P's four rows + the cell's two new rows (x <= 2, -x <= -1)
        if G.shape[0] == 6 and not hits and np.any(np.isclose(h, 2.0)) and np.any(np.isclose(h, -1.0)):
            hits.append(1)
            return dict(status=1, x=None, fun=None)
        return real(c, G, h, solver)

    monkeypatch.setattr(pc, "lpsolve", flaky)
    got = pc.region_diff(P.copy(), pc.Region([c.copy() for c in cells]))
    assert hits, "the injected failure was not reached"
    gp, wp = _pieces(pc, got), _pieces(pc, want)
    assert len(gp) == len(wp)
    for q, w in zip(gp, wp):
        assert np.allclose(q.A, w.A, atol=1e-12) and np.allclose(q.b, w.b, atol=1e-12)


def test_bounding_box_where_the_one_deviating_status_shows(pc):
    """g1 holds ONE LP (form F3, (5,3): min x_2 over an unbounded polytope) on which HiGHS's presolve reports status 2
    where a simplex finds 3 (tests/test_gpu_parity.py::test_lp_golden).  `bounding_box` of that polytope is the one
    caller where it shows (ref :1372-1384): the reference -- and this mirror on the scipy backend -- writes
    l[2] = 0 for status 2; the 'hip' backend, whose LP says "unbounded", writes -inf.  Both behaviours pinned here,
    every other entry of the box equal on both backends."""
    from polytope_amd import solvers
    g = load_golden("g1_lp.npz")
    dev = np.nonzero(g["status"] != g["status_nopresolve"])[0]
    assert len(dev) == 1
    i = int(dev[0])
    m, n = int(g["m"][i]), int(g["n"][i])
    assert (m, n) == (5, 3) and np.array_equal(g["c"][i, :n], [0.0, 0.0, 1.0])
    G, h = g["G"][i, :m * n].reshape(m, n), g["h"][i, :m]
    lo, hi = pc.bounding_box(pc.Polytope(G.copy(), h.copy(), normalize=False))
    if solvers.default_solver == "scipy":
        assert lo[2, 0] == 0.0                      # status 2 branch of the reference (:1380-1382)
    else:
        assert lo[2, 0] == -np.inf                  # status 3 branch (:1377-1379): the documented deviation
        solvers.default_solver = "scipy"
        try:
            lo_s, hi_s = pc.bounding_box(pc.Polytope(G.copy(), h.copy(), normalize=False))
        finally:
            solvers.default_solver = "hip"
        rest = np.ones((3, 1), bool)
        rest[2, 0] = False
        assert lo_s[2, 0] == 0.0
        assert np.array_equal(np.isinf(lo)[rest], np.isinf(lo_s)[rest]) and np.array_equal(np.isinf(hi), np.isinf(hi_s))
        fin = np.isfinite(hi_s)
        assert np.allclose(hi[fin], hi_s[fin], atol=TOL, rtol=0)
        fin = np.isfinite(lo_s) & rest
        assert np.allclose(lo[fin], lo_s[fin], atol=TOL, rtol=0)


def test_intersect_and_reduce_beyond_64_rows(pc, monkeypatch):
    """g14: Polytope.intersect stacks m1 + m2 rows (ref :268-275) and `reduce` takes any number (ref :1053-1163).  Same
    rows in the same order as the reference on both backends; on 'hip' the whole call is ONE fused launch
    (reduce_lds_kernel) -- the LP loop over single launches must not be reached."""
    from polytope_amd import solvers
    g = load_golden("g14_wide_reduce.npz")
    if solvers.default_solver == "hip":
        def boom(*a, **k):
            raise AssertionError("_reduce_lp_loop reached on the 'hip' backend")
        monkeypatch.setattr(pc, "_reduce_lp_loop", boom)
    for k in range(int(g["n"])):
        P = pc.Polytope(g["c%d_PA" % k], g["c%d_Pb" % k], normalize=False)
        if int(g["c%d_has_Q" % k]):
            Q = pc.Polytope(g["c%d_QA" % k], g["c%d_Qb" % k], normalize=False)
            R = P.intersect(Q)
        else:
            R = pc.reduce(P)
        assert R.A.shape == g["c%d_A" % k].shape, (k, R.A.shape)
        assert np.allclose(R.A, g["c%d_A" % k], atol=1e-9, rtol=0) and np.allclose(R.b, g["c%d_b" % k], atol=1e-9, rtol=0)
        assert bool(R.minrep) == bool(g["c%d_minrep" % k])
        assert abs(float(pc.cheby_ball(R)[0]) - float(g["c%d_r" % k])) <= TOL


def test_union_memos_do_not_change_results(pc):
    """union(check_convex=True) keeps the convexity verdict and the merged piece of a group by content (the repeated
    union of Region.intersect meets the same groups at every step): with the memos emptied before the call, filled by
    an earlier call, or filled by a DIFFERENT region, the pieces are the same, and a memoised piece is not aliased
    into a caller's edit."""
    import itertools
    shape = (4, 3, 2)
    cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(3)])
             for i in itertools.product(*[range(n) for n in shape])]
    rng = np.random.default_rng(3)
    A = rng.standard_normal((9, 3))
    A /= np.linalg.norm(A, axis=1)[:, None]
    P = pc.Polytope(A, 0.3 + A @ (0.5 * np.ones(3)))
    Q = pc.Polytope(A, 0.22 + A @ np.array([0.4, 0.6, 0.5]))

    def pieces(X):
        R = pc.Region([c.copy() for c in cells]).intersect(X.copy())
        return [(p.A.copy(), p.b.copy()) for p in (R.list_poly if isinstance(R, pc.Region) else [R])]

    def same(u, v):
        return len(u) == len(v) and all(np.array_equal(a0, a1) and np.array_equal(b0, b1) for (a0, b0), (a1, b1) in zip(u, v))
    pc._hull_memo.clear(); pc._convex_memo.clear()
    cold = pieces(P)
    warm = pieces(P)
    pieces(Q)                     # other groups in between
    again = pieces(P)
    pc._hull_memo.clear(); pc._convex_memo.clear()
    cold2 = pieces(P)
    assert len(cold) >= 1 and same(cold, warm) and same(cold, again) and same(cold, cold2)
    # a caller that edits a returned piece in place does not reach into later results
    R = pc.Region([c.copy() for c in cells]).intersect(P.copy())
    first = (R.list_poly if isinstance(R, pc.Region) else [R])[0]
    keepA = first.A.copy()
    first.A *= 2.0
    assert same(pieces(P), cold)
    first.A[...] = keepA


# ------------------------------------------------------------------ resident tables, cross pairs (hip only)
@pytest.mark.gpu
def test_cross_pairs_and_resident_tables():
    """plp_overlap_cross against the one-batch-per-member scan it replaces (the opening scan of region_diff, ref
    :2148-2158, for every (member, cell) pair), on boxes and random polytopes with ragged row counts, d = 2..9; then the
    packed tables: a Region's rows go to the device once (batch.h2d_bytes stands still on the second call)."""
    import itertools
    import polytope_amd.polytope as pc
    from polytope_amd import solvers, batch, prop2partition as p2p
    old = solvers.default_solver
    solvers.default_solver = "hip"
    try:
        rng = np.random.default_rng(5)
        for d, n1, n2 in [(2, 7, 40), (4, 30, 200), (6, 5, 33), (9, 4, 21)]:
            def rand_cell():
                lo = rng.random(d) * (2 if d <= 4 else 0.6)
                w = 0.3 + rng.random(d)
                box = pc.box2poly([[a, a + ww] for a, ww in zip(lo, w)])
                k = int(rng.integers(0, 4))
                if k == 0:
                    return box
                R = rng.standard_normal((k, d))
                R /= np.linalg.norm(R, axis=1)[:, None]
                c = lo + w / 2
                return pc.Polytope(np.vstack([box.A, R]), np.hstack([box.b, R @ c + 0.2 * rng.random(k)]))
            firsts = [rand_cell() for _ in range(n1)]
            seconds = [rand_cell() for _ in range(n2)]
            got = pc._cross_touch(firsts, seconds)
            want = np.array([[r >= pc.ABS_TOL for r in pc._radii_stacked(p, seconds)] for p in firsts])
            assert got.shape == (n1, n2) and np.array_equal(got, want), (d, np.argwhere(got != want)[:5])
            assert got.sum() < got.size and (d > 4 or got.sum() > 0)
        # mldivide / is_subset over the screening: the 4-D grid, 60 cells against all 200
        shape = (5, 5, 4, 2)
        cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)])
                 for i in itertools.product(*[range(n) for n in shape])]
        big = pc.Region(cells)
        assert pc.is_subset(pc.Region(cells[:60]), big)
        shifted = pc.Region([c.translation(np.array([0.9, 0.0, 0.0, 0.0])) for c in cells[:20]])
        assert not pc.is_subset(shifted, big)
        # resident rows: the second query uploads the points only; adjacency twice uploads nothing the second time
        pts = rng.random((4, 5000))
        inside = big.contains(pts)
        h0 = batch.h2d_bytes
        assert np.array_equal(big.contains(pts), inside) and inside.all()
        assert batch.h2d_bytes - h0 == pts.nbytes
        regs = [pc.Region([c]) for c in cells]
        adj = p2p.adjacency_matrix_dense(regs)
        h0 = batch.h2d_bytes
        assert np.array_equal(p2p.adjacency_matrix_dense(regs), adj)
        assert batch.h2d_bytes == h0
        regs[3] = pc.Region([cells[3].copy()])          # a member replaced: a new table
        assert np.array_equal(p2p.adjacency_matrix_dense(regs), adj) and batch.h2d_bytes > h0
        # rows edited IN PLACE (same arrays): the resident table must not answer for the old rows (ref :217-218 reads A, b per call)
        small = pc.Region(cells[:12])
        q = rng.random((4, 4000))
        before = small.contains(q)
        cells[0].b -= 0.05
        cells[5].A[:] = -cells[5].A
        after = small.contains(q)
        want = np.zeros(q.shape[1], dtype=bool)
        for c in cells[:12]:
            want |= np.all(c.A.dot(q) - c.b[:, None] < pc.ABS_TOL, axis=0)
        assert np.array_equal(after, want) and not np.array_equal(after, before)
    finally:
        solvers.default_solver = old


def test_remembered_tables_follow_their_lists():
    """(round-4 advisor) The flattened member list remembered on a partition / Region is keyed by EVERY element (an element
    replaced in the middle at unchanged length is seen), the packed table of a list is reused only while its content is
    what the members hold now (arrays edited in place get a new table), neither travels through pickle / deepcopy, and the
    quickhull rank screen leaves tiny-scale point sets to the reference's SVD test."""
    import copy
    import pickle
    import polytope_amd.polytope as pc
    from polytope_amd import prop2partition as p2p, quickhull as qh
    cells = [pc.box2poly([[i, i + 1.0], [0.0, 1.0]]) for i in range(5)]
    regs = [pc.Region([c]) for c in cells]
    owner = pc.Region(cells)
    m0, _, _ = p2p._flat_members(regs, owner)
    assert p2p._flat_members(regs, owner)[0] is m0                 # remembered
    new = pc.box2poly([[1.0, 2.0], [5.0, 6.0]])
    regs[2] = pc.Region([new])                                      # middle element replaced, same length
    m1, _, _ = p2p._flat_members(regs, owner)
    assert m1 is not m0 and m1[2] is new
    regs[1].list_poly.append(pc.box2poly([[7.0, 8.0], [0.0, 1.0]]))  # a member list grown inside an element
    m2, first, _ = p2p._flat_members(regs, owner)
    assert len(m2) == 6 and first.tolist() == [0, 1, 3, 4, 5, 6]
    # packed tables: same arrays, same content -> the same table; edited in place -> a new one with the new rows
    t0 = pc._table_of(cells, owner)
    assert pc._table_of(cells, owner) is t0
    cells[3].b += 0.25
    t1 = pc._table_of(cells, owner)
    assert t1 is not t0 and np.array_equal(t1.b[3], cells[3].b) and not np.array_equal(t0.b[3], cells[3].b)
    # no device state in copies: whatever the table holds on the device stays behind
    t1._dev = ("device tensors stand-in",)
    owner._p2p_flat = ("key", "value")
    for clone in (pickle.loads(pickle.dumps(owner)), copy.deepcopy(owner)):
        assert "_packed" not in clone.__dict__ and "_p2p_flat" not in clone.__dict__
        assert len(clone.list_poly) == 5 and np.array_equal(clone.list_poly[3].b, cells[3].b)
    tt = pickle.loads(pickle.dumps(t1))
    assert tt._dev is None and np.array_equal(tt.A, t1.A)
    part = p2p.Partition(pc.Region(cells))
    part._packed = ("k", t1)
    assert "_packed" not in pickle.loads(pickle.dumps(part)).__dict__
    # rank screen: a full-rank set at scale 1e-14 is NOT waved through (its singular values sit near the reference's 1e-15)
    rng = np.random.default_rng(0)
    P = rng.random((50, 3))
    assert qh._full_rank_clearly(P - P[0])
    assert not qh._full_rank_clearly((P - P[0]) * 1e-14)
