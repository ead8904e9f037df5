"""Differential test of the Python layer: every set operation on seeded random inputs once with
`solvers.default_solver = 'scipy'` (LPs one by one through scipy.optimize.linprog, exactly the reference's
own backend and call pattern) and once with `'hip'` (the batched device paths).  The two must agree:
booleans and piece counts exactly, rows / radii / boxes / vertices within 1e-9.

This is the check that the batching, packing and caching logic around the kernels (what
polytope_amd/polytope.py adds to the reference's algorithms) does not change a result.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def both(fn):
    """Run fn() under each backend -> (scipy result, hip result)."""
    from polytope_amd import solvers
    old = solvers.default_solver
    out = []
    try:
        for name in ("scipy", "hip"):
            solvers.default_solver = name
            out.append(fn())
    finally:
        solvers.default_solver = old
    return out


def rand_poly(pc, rng, d, m, centre=None, scale=1.0):
    G = rng.standard_normal((m, d))
    G /= np.linalg.norm(G, axis=1)[:, None]
    A = np.vstack([np.eye(d), -np.eye(d), G])
    b = np.r_[np.full(2 * d, 1.5 * scale), scale * (0.6 + 0.8 * rng.random(m))]
    if centre is not None:
        b = b + A @ centre
    return pc.Polytope(A, b)


def rows(p):
    Ab = np.c_[p.A, p.b]
    return Ab[np.lexsort(np.round(Ab, 7).T[::-1])]


def same_poly(p, q):
    return p.A.shape == q.A.shape and np.allclose(rows(p), rows(q), rtol=0, atol=TOL)


def same_region(r1, r2, pc):
    l1 = r1.list_poly if isinstance(r1, pc.Region) else ([r1] if r1.A.size else [])
    l2 = r2.list_poly if isinstance(r2, pc.Region) else ([r2] if r2.A.size else [])
    return len(l1) == len(l2) and all(same_poly(a, b) for a, b in zip(l1, l2))


@pytest.mark.parametrize("d", [2, 3])
def test_polytope_ops_agree(d):
    import polytope_amd.polytope as pc
    rng = np.random.default_rng(100 + d)
    for trial in range(6):
        c1 = 0.3 * rng.standard_normal(d)
        mk = lambda: (rand_poly(pc, np.random.default_rng(1000 * d + trial), d, 6),
                      rand_poly(pc, np.random.default_rng(2000 * d + trial), d, 5, centre=c1, scale=0.8))
        # reduce / cheby / bbox
        def basic():
            p, q = mk()
            pr = pc.reduce(p)
            return pr, pc.cheby_ball(q)[0], pc.bounding_box(q), pc.is_fulldim(p), pc.is_subset(q, p), pc.volume(p, 4000, seed=3)
        (s, h) = both(basic)
        assert same_poly(s[0], h[0]) and abs(s[1] - h[1]) <= TOL
        assert np.allclose(s[2][0], h[2][0], atol=TOL) and np.allclose(s[2][1], h[2][1], atol=TOL)
        assert s[3] == h[3] and s[4] == h[4] and abs(s[5] - h[5]) <= 1e-12
        # intersect / diff / union / envelope / adjacency
        def setops():
            p, q = mk()
            i = p.intersect(q)
            dreg = pc.mldivide(p, q)
            u = pc.union(p, q, check_convex=True)
            e = pc.envelope(pc.Region([p, q]))
            return i, dreg, u, e, pc.is_adjacent(p, q), pc.is_convex(pc.Region([p, q]))[0]
        (s, h) = both(setops)
        assert same_poly(s[0], h[0]), trial
        assert same_region(s[1], h[1], pc), trial
        assert same_region(s[2], h[2], pc), trial
        assert same_poly(s[3], h[3]), trial
        assert s[4] == h[4] and s[5] == h[5]


def test_region_ops_agree():
    import polytope_amd.polytope as pc
    from polytope_amd import prop2partition as p2p
    rng = np.random.default_rng(7)

    def build():
        r = np.random.default_rng(42)
        cells = [pc.box2poly([[i, i + 1.0], [j, j + 1.0]]) for i in range(3) for j in range(3)]
        blob = rand_poly(pc, r, 2, 7, centre=np.array([1.4, 1.6]), scale=0.9)
        other = rand_poly(pc, r, 2, 5, centre=np.array([2.1, 0.8]), scale=0.7)
        return cells, blob, other

    def ops():
        cells, blob, other = build()
        reg = pc.Region(cells)
        return (reg.intersect(blob), pc.region_diff(blob, pc.Region(cells[:5])), pc.mldivide(pc.Region([blob, other]), cells[4]),
                p2p.find_adjacent_regions(cells).toarray(), p2p.are_disjoint(cells + [blob]),
                pc.is_subset(pc.Region([blob]), reg), reg.contains(np.array([[0.5, 1.5, 2.9, 3.1], [0.5, 2.5, 2.9, 0.2]])),
                pc.bounding_box(pc.Region([blob, other])))
    (s, h) = both(ops)
    assert same_region(s[0], h[0], pc) and same_region(s[1], h[1], pc) and same_region(s[2], h[2], pc)
    assert np.array_equal(s[3], h[3]) and s[4] == h[4] and s[5] == h[5] and np.array_equal(s[6], h[6])
    assert np.allclose(s[7][0], h[7][0], atol=TOL) and np.allclose(s[7][1], h[7][1], atol=TOL)


@pytest.mark.parametrize("d", [2, 3, 4])
def test_vertex_enumeration_agrees(d):
    import polytope_amd.polytope as pc

    def ops():
        p = rand_poly(pc, np.random.default_rng(50 + d), d, 7)
        np.random.seed(d)
        V = pc.extreme(p)
        np.random.seed(d + 1)
        q = pc.qhull(np.random.default_rng(d).standard_normal((60, d)))
        return V[np.lexsort(np.round(V, 7).T[::-1])], q
    (s, h) = both(ops)
    assert s[0].shape == h[0].shape and np.allclose(s[0], h[0], rtol=0, atol=TOL)
    assert np.array_equal(s[1].A, h[1].A) and np.allclose(s[1].b, h[1].b, rtol=0, atol=1e-12)
