"""ctypes wrapper of oracle/libplp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (polytope_amd/) never does.

Each function mirrors one C entry point of plp_oracle.c, which cites the reference
file:line it restates (polytope/solvers.py, polytope/polytope.py, polytope/quickhull.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libplp_oracle.so")

RF_EMPTY, RF_EARLY, RF_MINREP, RF_LPFAIL = 1, 2, 4, 8
RF_F1OPEN = 32   # RF_EMPTY because the Chebyshev LP did not end optimal (polytope_amd/csrc/plp_common.hpp)


def build(force=False):
    """Compile the C restatement with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "plp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.plpo_lp_solve.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, ip]
        L.plpo_lp_solve.restype = C.c_int
        L.plpo_cheby.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, ip]
        L.plpo_cheby.restype = C.c_int
        L.plpo_bounding_box.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, ip]
        L.plpo_bounding_box.restype = C.c_int
        L.plpo_reduce.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double,
                                  C.POINTER(C.c_uint64), dp, dp, dp, ip]
        L.plpo_reduce.restype = C.c_int
        L.plpo_reduce_batch.argtypes = [C.c_int64, C.c_int, C.c_int, dp, dp, C.c_double, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_int32), dp, C.POINTER(C.c_int32)]
        L.plpo_reduce_batch.restype = C.c_int
        L.plpo_contains.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int32),
                                    C.c_int64, dp, C.c_double, C.POINTER(C.c_uint8)]
        L.plpo_contains.restype = None
        L.plpo_region_contains.argtypes = L.plpo_contains.argtypes
        L.plpo_region_contains.restype = None
        L.plpo_assign.argtypes = [C.c_int64, C.c_int, dp, C.c_int, dp, dp, C.c_double,
                                  C.POINTER(C.c_int32), dp, C.POINTER(C.c_int64), dp]
        L.plpo_assign.restype = None
        L.plpo_hull_reassign.argtypes = [C.c_int64, C.c_int, dp, C.POINTER(C.c_int32), dp, C.POINTER(C.c_uint8),
                                         C.c_int, C.c_int, dp, dp, C.c_double, C.POINTER(C.c_int64), dp,
                                         C.POINTER(C.c_int64)]
        L.plpo_hull_reassign.restype = None
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def lp_solve(c, G, h):
    """min c'x s.t. Gx<=h, x free -> (status, x or None, fun or None, iters).  solvers.py:149-158."""
    c, G, h = _d(c).ravel(), _d(G), _d(h).ravel()
    n = c.size
    G = G.reshape(-1, n)
    m = G.shape[0]
    x = np.empty(n)
    fun = C.c_double()
    it = C.c_int()
    st = lib().plpo_lp_solve(m, n, _p(c), _p(G), _p(h), _p(x), C.byref(fun), C.byref(it))
    if st != 0:
        return st, None, None, it.value
    return st, x, fun.value, it.value


def cheby(A, b):
    """Raw F1 LP of cheby_ball (polytope.py:1283-1288) -> (status, r, xc)."""
    A, b = _d(A), _d(b).ravel()
    m, d = A.shape
    r = C.c_double()
    xc = np.empty(d)
    st = lib().plpo_cheby(m, d, _p(A), _p(b), C.byref(r), _p(xc), None)
    return st, r.value, xc


def cheby_ball(A, b):
    """cheby_ball semantics (polytope.py:1289-1300): (r, xc) or (0.0, None)."""
    st, r, xc = cheby(A, b)
    if st != 0 or r < 0:
        return 0.0, None
    return r, xc


def bounding_box(A, b):
    """polytope.py:1367-1409 -> (lb, ub, bad_status)."""
    A, b = _d(A), _d(b).ravel()
    m, d = A.shape
    lb, ub = np.empty(d), np.empty(d)
    nlp = C.c_int(0)
    bad = lib().plpo_bounding_box(m, d, _p(A), _p(b), _p(lb), _p(ub), C.byref(nlp))
    return lb, ub, bad


def reduce(A, b, abs_tol=1e-7):
    """polytope.py:1053-1163 on one non-minrep polytope.

    -> dict(keep=bool[m], flags, b=b values after the 0.1 round trip, r, xc, nlp)
    """
    A, b = _d(A), _d(b).ravel()
    m, d = A.shape
    keep = (C.c_uint64 * 4)()   # PLPO_KEEP_WORDS words: polytopes of up to 256 rows
    bout = np.empty(max(m, 1))
    r = C.c_double()
    xc = np.empty(d)
    nlp = C.c_int()
    if m > 256:
        raise ValueError("oracle.reduce: up to 256 rows (PLPO_MAXM)")
    flags = lib().plpo_reduce(m, d, _p(A), _p(b), abs_tol, keep, _p(bout),
                              C.byref(r), _p(xc), C.byref(nlp))
    mask = np.array([(keep[i >> 6] >> (i & 63)) & 1 for i in range(m)], dtype=bool)
    return dict(keep=mask, mask=int(keep[0]), words=[int(w) for w in keep], flags=flags, b=bout[:m], r=r.value, xc=xc,
                nlp=nlp.value)


def reduce_batch(A, b, abs_tol=1e-7):
    """plpo_reduce on every polytope of a packed batch A[B,m,d], b[B,m] (m <= 64), the loop in C.

    -> dict(keep=uint64[B] (bit i <=> input row i kept), flags=int32[B], r=f64[B], nlp=int32[B])
    """
    A, b = _d(A), _d(b)
    B, m, d = A.shape
    keep = np.empty(B, dtype=np.uint64)
    flags = np.empty(B, dtype=np.int32)
    nlp = np.empty(B, dtype=np.int32)
    r = np.empty(B)
    rc = lib().plpo_reduce_batch(B, m, d, _p(A), _p(b), abs_tol, _p(keep, C.c_uint64), _p(flags, C.c_int32), _p(r),
                                 _p(nlp, C.c_int32))
    if rc != 0:
        raise ValueError("oracle.reduce_batch: m <= 64, d <= 16")
    return dict(keep=keep, flags=flags, r=r, nlp=nlp)


def contains(A, b, X, abs_tol=1e-7, mrows=None, region=False):
    """A:[P][m_max][d], b:[P][m_max], X:[N][d] point-major.

    region=False -> uint8 [P][N]  (Polytope.contains, polytope.py:217-218)
    region=True  -> uint8 [N]     (Region.contains, polytope.py:732-746)
    """
    A, b, X = _d(A), _d(b), _d(X)
    P, m_max, d = A.shape
    N = X.shape[0]
    mr = None if mrows is None else np.ascontiguousarray(mrows, dtype=np.int32)
    mp = None if mr is None else _p(mr, C.c_int32)
    if region:
        out = np.empty(N, dtype=np.uint8)
        lib().plpo_region_contains(P, m_max, d, _p(A), _p(b), mp, N, _p(X), abs_tol, _p(out, C.c_uint8))
    else:
        out = np.empty((P, N), dtype=np.uint8)
        lib().plpo_contains(P, m_max, d, _p(A), _p(b), mp, N, _p(X), abs_tol, _p(out, C.c_uint8))
    return out


def assign(X, normals, offsets, abs_tol=1e-7):
    """quickhull outside-set assignment + furthest (quickhull.py:87-102,117-121,224-245).

    -> (facet_of_point int32[N], dist f64[N], argmax int64[F], max f64[F])
    """
    X, normals, offsets = _d(X), _d(normals), _d(offsets).ravel()
    N, d = X.shape
    F = normals.shape[0]
    fop = np.empty(N, dtype=np.int32)
    dist = np.empty(N)
    am = np.empty(F, dtype=np.int64)
    mx = np.empty(F)
    lib().plpo_assign(N, d, _p(X), F, _p(normals), _p(offsets), abs_tol,
                      _p(fop, C.c_int32), _p(dist), _p(am, C.c_int64), _p(mx))
    return fop, dist, am, mx


def hull_reassign(X, owner, dist, dead, new_id0, normals, offsets, abs_tol=1e-7):
    """One quickhull outside-set update (quickhull.py:273-283, :311-336, :87-102); owner/dist in place.

    -> (count int64[n_new], argmax int64[n_new], maxd f64[n_new])
    """
    X, normals, offsets = _d(X), _d(normals), _d(offsets).ravel()
    N, d = X.shape
    n_new = normals.shape[0]
    assert owner.dtype == np.int32 and dist.dtype == np.float64 and dead.dtype == np.uint8
    am = np.empty(n_new, dtype=np.int64)
    mx = np.empty(n_new)
    cnt = np.empty(n_new, dtype=np.int64)
    lib().plpo_hull_reassign(N, d, _p(X), _p(owner, C.c_int32), _p(dist), _p(dead, C.c_uint8), int(new_id0), n_new,
                             _p(normals), _p(offsets), abs_tol, _p(am, C.c_int64), _p(mx), _p(cnt, C.c_int64))
    return cnt, am, mx


class HullSession:
    """CPU stand-in with the interface of polytope_amd.batch.HullSession (tests drive the host-side
    quickhull logic with it where no GPU is present, and compare the two step by step on the GPU)."""

    def __init__(self, X):
        self.X = _d(X)
        self.N, self.d = self.X.shape
        self.owner = np.zeros(self.N, np.int32)
        self.dist = np.zeros(self.N)
        self.dead = np.zeros(4096, np.uint8)
        self.next_id = 1

    def drop(self, idx):
        self.owner[np.asarray(idx, dtype=np.int64)] = -1

    def reassign(self, dead_ids, normals, offsets, abs_tol=1e-7):
        normals = _d(normals).reshape(-1, self.d)
        n_new = normals.shape[0]
        while self.dead.size < self.next_id + n_new:
            self.dead = np.concatenate([self.dead, np.zeros_like(self.dead)])
        dead_ids = np.asarray(dead_ids, dtype=np.int64).ravel()
        assert np.all(dead_ids >= 0) and np.all(dead_ids < self.next_id), "dead facet id was never handed out"
        self.dead[dead_ids] = 1
        id0 = self.next_id
        cnt, am, mx = hull_reassign(self.X, self.owner, self.dist, self.dead, id0, normals, offsets, abs_tol)
        self.next_id += n_new
        return id0, cnt, am, mx

    def read(self):
        return self.owner.copy(), self.dist.copy()

    def close(self):
        pass
