"""CPU: the one-LP-per-lane engine of the fused reduce (polytope_amd/csrc/plp_lane_lp.hpp, used by plp_reduce_lane.hip for
the box and redundancy LPs at d <= 3) compiled for the HOST and run against the oracle's dictionary simplex on every
F3 / F2 LP of random, unbounded and structured polytopes: same status, optimum within 1e-11 (relative beyond 1), and nothing handed back on
random data.  The header is the same source the device compiles (explicit fma, -ffp-contract=off on both sides)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def lane(oracle, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("lane") / "liblane_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(ROOT, "tests", "cabi", "lane_lp_host.cpp"), "-L", os.path.join(ROOT, "oracle"),
                           "-lplp_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(out)
    dp = C.POINTER(C.c_double)
    L.lane_check.argtypes = [C.c_longlong, dp, dp, C.POINTER(C.c_int), dp, C.c_int]
    L.lane_check4.argtypes = L.lane_check.argtypes
    L.lane_solve_one.argtypes = [dp, dp, dp, dp, C.POINTER(C.c_int)]
    L.lane_solve_one4m.argtypes = [C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int)]

    def check(A, b, m=None, which=3):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        B, mm, d = A.shape
        assert d in (3, 4)
        fn = L.lane_check if d == 3 else L.lane_check4
        if mm < 16:
            A = np.concatenate([A, np.zeros((B, 16 - mm, d))], axis=1)
            b = np.concatenate([b, np.zeros((B, 16 - mm))], axis=1)
            m = np.full(B, mm, np.int32) if m is None else m
        st = np.zeros(32)
        mp = None if m is None else np.ascontiguousarray(m, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
        fn(B, A.ctypes.data_as(dp), b.ctypes.data_as(dp), mp, st.ctypes.data_as(dp), which)
        return dict(lps=int(st[0]), retry=int(st[1]), status_diff=int(st[2]), opt=int(st[3]), unb=int(st[4]),
                    max_diff=float(st[5]), mean_iters=st[6] / max(st[0], 1), max_iters=int(st[7]))
    def solve_one(A16, beta16, c):
        """one LP  min c.x  s.t.  A16 x <= beta16  from x = 0 through walk3: (status, x)"""
        x = np.zeros(3)
        it = C.c_int(0)
        st = L.lane_solve_one(np.ascontiguousarray(A16).ctypes.data_as(dp), np.ascontiguousarray(beta16).ctypes.data_as(dp),
                              np.ascontiguousarray(c, dtype=np.float64).ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(it))
        return st, x
    check.solve_one = solve_one

    def solve_one4(A, beta, c):
        """the same in R^4 through walk4, any number of row slots: (status, x)"""
        x = np.zeros(4)
        it = C.c_int(0)
        A = np.ascontiguousarray(A, dtype=np.float64)
        st = L.lane_solve_one4m(A.shape[0], A.ctypes.data_as(dp), np.ascontiguousarray(beta, dtype=np.float64).ctypes.data_as(dp),
                                np.ascontiguousarray(c, dtype=np.float64).ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(it))
        return st, x
    check.solve_one4 = solve_one4
    return check


def test_lane_engine_equals_the_dictionary_simplex_on_random_polytopes(lane):
    from polytope_amd.synth import random_hpolytopes
    tot = 0
    for seed, m in [(0, 16), (1, 16), (2, 12), (3, 7)]:
        A, b = random_hpolytopes(6000, m, 3, seed=seed)
        s = lane(A, b)
        assert s["status_diff"] == 0 and s["retry"] == 0 and s["max_diff"] <= 1e-11, s
        assert s["opt"] == s["lps"] and s["max_iters"] <= 12 and 1.0 < s["mean_iters"] < 3.5, s
        tot += s["lps"]
    assert tot > 400000


def test_lane_engine_unbounded_and_ragged(lane):
    from polytope_amd.synth import random_hpolytopes
    A, b = random_hpolytopes(6000, 16, 3, seed=5, bounded=False)
    m = np.random.default_rng(0).integers(1, 17, size=6000).astype(np.int32)
    for k in range(6000):
        A[k, m[k]:] = 0.0
        b[k, m[k]:] = 0.0
    s = lane(A, b, m)
    assert s["status_diff"] == 0 and s["max_diff"] <= 1e-11, s
    assert s["retry"] <= s["lps"] // 1000, s
    # polytopes with a bounded Chebyshev ball that are unbounded themselves: prisms open along a random axis
    rng = np.random.default_rng(2)
    A2 = np.zeros((500, 16, 3))
    b2 = np.zeros((500, 16))
    for k in range(500):
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.random(n) * 2 * np.pi)
        A2[k, :n] = np.cos(ang)[:, None] * Q[0] + np.sin(ang)[:, None] * Q[1]   # normals in a plane: open along Q[2]
        b2[k, :n] = 0.5 + rng.random(n)
        if k % 2:                                                                  # half of them closed on one side
            A2[k, n], b2[k, n] = Q[2], 1.0
            n += 1
        b2[k, n:] = 0.0
    s = lane(A2, b2, np.array([int(np.count_nonzero(np.abs(A2[k]).sum(axis=1))) for k in range(500)], np.int32))
    assert s["status_diff"] == 0 and s["max_diff"] <= 1e-11 and s["unb"] > 300 and s["opt"] > 300, s


def test_lane_engine_structured_polytopes(lane):
    """Ties in the ratio test, degenerate vertices, duplicated / tangent rows, lattice normals (tests/structured_cases.py):
    what is not handed back agrees with the simplex; what is handed back stays a small share (it costs a second pass)."""
    from structured_cases import structured_polytopes
    A, b, fam = structured_polytopes(4096)
    s = lane(A, b)
    assert s["status_diff"] == 0 and s["max_diff"] <= 1e-9, s
    assert s["retry"] <= 0.05 * s["lps"], s
    # boxes and prisms: the LPs end on a facet or an edge (multipliers of fewer than three active rows)
    box = np.vstack([np.eye(3), -np.eye(3)])
    A2 = np.zeros((64, 16, 3))
    b2 = np.zeros((64, 16))
    rng = np.random.default_rng(3)
    for k in range(64):
        lo = rng.integers(-3, 3, 3) * 0.5
        hi = lo + rng.choice([0.5, 1.0, 2.0], 3)
        A2[k, :6], b2[k, :6] = box, np.r_[hi, -lo]
        n = np.array([[1, 1, 0], [0, 1, 1], [1, 0, -1], [1, 1, 1]], float)
        n /= np.linalg.norm(n, axis=1)[:, None]
        A2[k, 6:10] = n
        b2[k, 6:10] = n @ ((lo + hi) / 2) + rng.choice([0.2, 0.5, 5.0], 4)
    s = lane(A2, b2, np.full(64, 10, np.int32))
    assert s["status_diff"] == 0 and s["max_diff"] <= 1e-11 and s["retry"] <= 0.05 * s["lps"], s


def test_walk_in_r4_equals_the_dictionary_simplex(lane):
    """walk4 (d = 4: projections by the Gram matrix of the active rows and its adjugate, the vertex test by generalised
    cross products): every box and redundancy LP of random (16,4), (12,4), (9,4) polytopes, bounded and not."""
    from polytope_amd.synth import random_hpolytopes
    for seed, m, bounded in [(0, 16, True), (1, 12, True), (2, 16, False), (3, 9, True)]:
        A, b = random_hpolytopes(5000, m, 4, seed=seed, bounded=bounded)
        s = lane(A, b)
        assert s["status_diff"] == 0 and s["max_diff"] <= 1e-10 and s["retry"] <= s["lps"] // 5000, s
        assert s["opt"] + s["unb"] + s["retry"] == s["lps"] and s["max_iters"] <= 16, s
    # boxes with cuts through their corners / along their edges: walks that end on a facet, an edge or a 2-face
    rng = np.random.default_rng(8)
    box = np.vstack([np.eye(4), -np.eye(4)])
    A2 = np.zeros((200, 16, 4))
    b2 = np.zeros((200, 16))
    for k in range(200):
        lo = rng.integers(-3, 3, 4) * 0.5
        hi = lo + rng.choice([0.5, 1.0, 2.0], 4)
        A2[k, :8], b2[k, :8] = box, np.r_[hi, -lo]
        n = np.array([[1, 1, 0, 0], [0, 1, 1, 0], [1, 0, 0, -1], [1, 1, 1, 1], [0, 0, 1, -1]], float)
        n /= np.linalg.norm(n, axis=1)[:, None]
        A2[k, 8:13] = n
        b2[k, 8:13] = n @ ((lo + hi) / 2) + rng.choice([0.2, 0.5, 5.0], 5)
    s = lane(A2, b2, np.full(200, 13, np.int32))
    assert s["status_diff"] == 0 and s["max_diff"] <= 1e-10 and s["retry"] <= 0.05 * s["lps"], s


def test_walk_stays_on_its_planes_when_the_cost_is_nearly_a_row_normal(lane, oracle):
    """Box LPs of a cube with copies of its rows tilted by 1e-9 .. 1e-4 (found by scripts/soak_lane.py): the projection of
    the cost onto such a row's plane is the small difference of two long vectors, and before it was orthogonalised a second
    time the walk left the plane by 5e-9 .. 3e-8 over a step.  The reference here is EXACT: every vertex of the polytope
    (all row triples, solved in extended precision), the best feasible one -- the dictionary simplex, the oracle's too,
    stops at reduced costs below its absolute 1e-9 and is itself 6e-9 off on some of these."""
    import itertools
    LD = np.longdouble
    rng = np.random.default_rng(3)
    box = np.vstack([np.eye(3), -np.eye(3)])

    def det3(M):
        return (M[0, 0] * (M[1, 1] * M[2, 2] - M[1, 2] * M[2, 1]) - M[0, 1] * (M[1, 0] * M[2, 2] - M[1, 2] * M[2, 0])
                + M[0, 2] * (M[1, 0] * M[2, 1] - M[1, 1] * M[2, 0]))

    worst = 0.0
    solved = handed = 0
    for k in range(60):
        A = np.zeros((16, 3))
        b = np.zeros(16)
        A[:6], b[:6] = box, 3.0
        for j in range(6, 9):
            i = int(rng.integers(0, 6))
            A[j] = A[i] + rng.choice([1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4]) * rng.standard_normal(3)
            b[j] = b[i] + rng.choice([0.0, 0.0, 1e-7, -1e-7])
        so, r, xc = oracle.cheby(A[:9], b[:9])
        assert so == 0 and r > 1.0
        beta = np.zeros(16)
        beta[:9] = np.maximum(b[:9] - A[:9] @ xc, 0.0)
        Al, bl = A[:9].astype(LD), beta[:9].astype(LD)
        verts = []
        for t in itertools.combinations(range(9), 3):
            M = Al[list(t)]
            D = det3(M)
            if abs(D) < 1e-14:   # (pairs of tilted copies: no vertex worth the name; the optimum is never only there)
                continue
            x = np.zeros(3, LD)
            for q in range(3):
                Mq = M.copy()
                Mq[:, q] = bl[list(t)]
                x[q] = det3(Mq) / D
            if np.all(Al @ x - bl <= 1e-13):
                verts.append(x)
        verts = np.array(verts)
        for it in range(6):
            c = np.zeros(3)
            c[it >> 1] = -1.0 if it & 1 else 1.0
            st, x = lane.solve_one(A, beta, c)
            if st == 5:
                handed += 1
                continue
            assert st == 0, (k, it, st)
            exact = float((verts @ c.astype(LD)).min())
            worst = max(worst, abs(float(c @ x) - exact))
            solved += 1
    assert worst <= 1e-12 and solved >= 300 and handed <= 60, (worst, solved, handed)


def test_walk4_stays_on_its_planes_when_the_cost_is_nearly_in_their_span(lane, oracle):
    """Nine (23,4) polytopes of scripts/soak_lane.py 60 23 (trial 47, family `dup`: row 9 is a copy of the box row +e_1 tilted by
    1e-9).  With two or three active rows that nearly contain the cost, walk4's direction `sum m_j n_j - det c` was what
    rounding left of two long vectors; a step of 1.4e12 times its length left the planes by 1.4e-4, the box value upper_1 came
    out 1.4e-4 short, the prefilter (-1e-4) dropped a facet: keep masks and LP counts off on 9 of 32 388 polytopes.  Now: three
    active rows -> the line gcross(n_0, n_1, n_2); two -> projected a second time where little of the cost is left."""
    from conftest import load_golden
    g = load_golden("lane_w4_tilted.npz")
    worst = 0.0
    for A, b, m in zip(g["A"], g["b"], g["m"]):
        A, b = A[:m].copy(), b[:m].copy()
        nrm = np.sqrt((A * A).sum(1))
        An, bn = A / nrm[:, None], b / nrm
        live = np.ones(m, bool)
        for i in range(m):            # the dedupe of reduce (ref :1094-1110), as the kernel applies it before the box LPs
            for j in range(i + 1, m):
                if An[i] @ An[j] > 1 - 1e-7:
                    live[j if bn[i] < bn[j] else i] = False
        so, r, xc = oracle.cheby(A, b)
        assert so == 0
        Ad = np.where(live[:, None], A, 0.0)
        beta = np.where(live, np.maximum(b - A @ xc, 0.0), 0.0)
        lo, hi, bad = oracle.bounding_box(A, b)
        assert not bad
        for it in range(8):
            c = np.zeros(4)
            c[it >> 1] = -1.0 if it & 1 else 1.0
            st, x = lane.solve_one4(Ad, beta, c)
            assert st == 0, (it, st)
            assert np.all(Ad @ x - beta <= 1e-9)                         # on the polytope
            worst = max(worst, abs(xc[it >> 1] + x[it >> 1] - (hi if it & 1 else lo)[it >> 1]))
    # (the oracle's own simplex stops at reduced costs below an absolute 1e-9: it is the looser side on these rows)
    assert worst <= 5e-9, worst


def test_walk_in_r4_on_rows_of_very_different_lengths(lane, oracle):
    """Rows scaled by e^-2.5 .. e^2.5 (scripts/soak_lane.py, family `scaled`) in R^4.  Found by the soak (seed 112, a (19,4)
    polytope, tests/golden/found/lane112_t11_k25610.npz): three active rows of norms 2.3 / 0.1 / 2.2, det of their Gram matrix
    2.7e-6, and the cancellation residue of a zero multiplier (1.6e-16) fell below -LANE_TOL_D det: the row was dropped, the
    direction on the two that stayed -- the cost parallel to one of them -- was rounding noise, and the walk followed it for
    t = 1.8e15: box value +0.02 where the optimum is -1.26, two facets removed by the prefilter."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import soak_lane as SL
    z = np.load(os.path.join(root, "tests", "golden", "found", "lane112_t11_k25610.npz"))
    A, b = z["A"], z["b"]
    st, r, xc = oracle.cheby(A, b)
    assert st == 0 and r > 0.5
    beta = np.maximum(b - A @ xc, 0.0)
    for k in range(4):
        for sgn in (1.0, -1.0):
            c = np.zeros(4)
            c[k] = sgn
            sw, x = lane.solve_one4(A, beta, c)
            so, xo, fo, _ = oracle.lp_solve(c, A, beta)
            assert sw == so == 0 and abs(float(c @ x) - fo) <= 1e-10 * max(1.0, abs(fo)), (k, sgn, sw, float(c @ x), so, fo)
    rng = np.random.default_rng(12)
    for m in (9, 12, 16):
        A, b, _ = SL.make(rng, 4000, m, 4, "scaled")
        s = lane(A, b)
        assert s["status_diff"] == 0 and s["max_diff"] <= 1e-10 and s["retry"] <= s["lps"] // 2000 + 2, s


def test_walk_in_r3_on_half_infinite_prisms(lane, oracle):
    """walk3 (the bench kernel's engine) where walk4 failed (above): unbounded polytopes -- half-infinite prisms -- with rows
    scaled by e^-2.5 .. e^2.5 and near-copies of rows (nearly degenerate vertices), costs orthogonal to the unbounded edge (a finite
    optimum attained along a ray: a multiplier that is zero up to rounding must not send the walk down that ray as "unbounded")
    and general ones.  Status equal to the oracle's on every LP, optimum within 1e-10."""
    rng = np.random.default_rng(31)
    n_lp = 0
    for trial in range(900):
        u = rng.standard_normal(3)
        u /= np.linalg.norm(u)
        m = int(rng.integers(5, 16))
        N = rng.standard_normal((m, 3))
        N -= np.outer(N @ u, u)
        N /= np.linalg.norm(N, axis=1)[:, None]
        b = 0.5 + rng.random(m)
        N[0], b[0] = -u, 1.0
        if trial % 3 >= 1:
            N *= np.exp(rng.uniform(-2.5, 2.5, m))[:, None]
            b *= np.linalg.norm(N, axis=1)
        if trial % 3 == 2:
            N[rng.integers(1, m)] = N[rng.integers(1, m)] * (1 + 1e-3 * rng.standard_normal())
        A16, b16 = np.zeros((16, 3)), np.zeros(16)
        A16[:m], b16[:m] = N, b
        for rep in range(6):
            c = rng.standard_normal(3)
            if rep < 4:
                c -= (c @ u) * u
            sw, x = lane.solve_one(A16, b16, c)
            so, xo, fo, _ = oracle.lp_solve(c, N, b)
            n_lp += 1
            if sw == 5:      # handed back
                continue
            assert sw == so, (trial, rep, sw, so)
            if sw == 0:
                assert abs(float(c @ x) - fo) <= 1e-10 * max(1.0, abs(fo)), (trial, rep, float(c @ x), fo)
    assert n_lp == 5400
