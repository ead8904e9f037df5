"""Batched entry points of the MI355X engine (no counterpart in the reference, which loops
in Python: polytope/polytope.py:1142-1151, :1367-1409, :2148-2152, prop2partition.py:57-61).

Every function takes either numpy arrays (host path: the C ABI copies in and out and
blocks) or torch CUDA tensors (device path: pointers are handed to the `_dev` entry points
on torch's current stream, results come back as CUDA tensors, nothing synchronises).

Packing: A[B, m_max, d], b[B, m_max], optional int32 m[B] (rows used per polytope).
"""
import ctypes as C

import numpy as np

from . import _lib

MAX_M, MAX_D = 64, 16


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _finite_or_raise(what, *arrays):
    for a in arrays:
        if a is not None and not np.all(np.isfinite(a)):
            # same exception class scipy.optimize.linprog raises on inf/nan input
            raise ValueError("%s: input must not contain values inf, nan, or None" % what)


# Debug counter: bytes the host-pointer entry points of this module handed to the library for upload (and the bytes
# polytope_amd.polytope moved to the device when it made a packed table resident).  Device-pointer calls add nothing.
h2d_bytes = 0


def _count_h2d(*arrays):
    global h2d_bytes
    for a in arrays:
        if a is not None:
            h2d_bytes += int(a.nbytes)


def _torch_stream_ctx(t):
    import torch
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    ctx = _lib.context(dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    return torch, ctx, stream


def _tprep(torch, t, dtype):
    if t is None:
        return None
    if t.dtype != dtype or not t.is_contiguous():
        t = t.to(dtype).contiguous()
    return t


# --------------------------------------------------------------------------------------
def lpsolve_batch(c, G, h, m=None):
    """B independent LPs  min c'x s.t. Gx <= h, x free  (solvers.py:76-106 semantics per LP).

    c[B,n], G[B,m_max,n], h[B,m_max] -> dict(status int32[B], x[B,n], fun[B], iters int32[B]);
    x/fun are NaN where status != 0.
    """
    lib = _lib.load()
    if _is_torch(G):
        torch, ctx, stream = _torch_stream_ctx(G)
        G = _tprep(torch, G, torch.float64)
        c = _tprep(torch, c, torch.float64)
        h = _tprep(torch, h, torch.float64)
        m = _tprep(torch, m, torch.int32)
        B, m_max, n = G.shape
        x = torch.empty((B, n), dtype=torch.float64, device=G.device)
        fun = torch.empty((B,), dtype=torch.float64, device=G.device)
        status = torch.empty((B,), dtype=torch.int32, device=G.device)
        iters = torch.empty((B,), dtype=torch.int32, device=G.device)
        _lib.check(lib.plp_lp_solve_batch_dev(ctx.handle, stream, B, m_max, n, _ptr(c), _ptr(G), _ptr(h), _ptr(m),
                                              _ptr(x), _ptr(fun), _ptr(status), _ptr(iters)), "plp_lp_solve_batch_dev")
        return dict(status=status, x=x, fun=fun, iters=iters)
    G = _np(G)
    if G.ndim != 3:
        raise ValueError("G must be [B, m_max, n]")
    B, m_max, n = G.shape
    c = _np(c).reshape(B, n)
    h = _np(h).reshape(B, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(B)
    # (inf / nan in the inputs: ValueError from the library, which checks them while staging -- plp_ctx_set_check_finite)
    x = np.empty((B, n))
    fun = np.empty(B)
    status = np.empty(B, np.int32)
    iters = np.empty(B, np.int32)
    _count_h2d(c, G, h, mm)
    _lib.check(lib.plp_lp_solve_batch(_lib.context().handle, B, m_max, n, _ptr(c), _ptr(G), _ptr(h), _ptr(mm),
                                      _ptr(x), _ptr(fun), _ptr(status), _ptr(iters)), "plp_lp_solve_batch")
    return dict(status=status, x=x, fun=fun, iters=iters)


def cheby_ball_batch(A, b, m=None):
    """Chebyshev-ball LP (F1, polytope.py:1283-1288) of B polytopes.

    -> dict(r[B] raw x[-1], xc[B,d], status[B]).  cheby_ball's own post-processing
    (status != 0 or r < 0 -> (0, None), :1289-1297) is left to the caller.
    """
    lib = _lib.load()
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        B, m_max, d = A.shape
        r = torch.empty((B,), dtype=torch.float64, device=A.device)
        xc = torch.empty((B, d), dtype=torch.float64, device=A.device)
        status = torch.empty((B,), dtype=torch.int32, device=A.device)
        _lib.check(lib.plp_cheby_batch_dev(ctx.handle, stream, B, m_max, d, _ptr(A), _ptr(b), _ptr(m), _ptr(r),
                                           _ptr(xc), _ptr(status)), "plp_cheby_batch_dev")
        return dict(r=r, xc=xc, status=status)
    A = _np(A)
    if A.ndim != 3:
        raise ValueError("A must be [B, m_max, d]")
    B, m_max, d = A.shape
    b = _np(b).reshape(B, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(B)
    # (inf / nan in the inputs: ValueError from the library, which checks them while staging -- plp_ctx_set_check_finite)
    r = np.empty(B)
    xc = np.empty((B, d))
    status = np.empty(B, np.int32)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_cheby_batch(_lib.context().handle, B, m_max, d, _ptr(A), _ptr(b), _ptr(mm), _ptr(r),
                                   _ptr(xc), _ptr(status)), "plp_cheby_batch")
    return dict(r=r, xc=xc, status=status)


def bbox_batch(A, b, m=None):
    """Bounding boxes of B polytopes with 1 <= d <= 16 (bounding_box's LP loops, polytope.py:1367-1409): the
    Chebyshev LP and 2d LPs from its centre per polytope, one launch.

    -> dict(lb[B,d], ub[B,d], status[B]): status 0 = box valid (+-inf where unbounded), 1 = polytope not handled
    here (no centre with r >= 1e-6, or a Bland case): the caller solves the 2d generic LPs for it.
    """
    lib = _lib.load()
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        B, m_max, d = A.shape
        lb = torch.empty((B, d), dtype=torch.float64, device=A.device)
        ub = torch.empty((B, d), dtype=torch.float64, device=A.device)
        status = torch.empty((B,), dtype=torch.int32, device=A.device)
        _lib.check(lib.plp_bbox_batch_dev(ctx.handle, stream, B, m_max, d, _ptr(A), _ptr(b), _ptr(m), _ptr(lb),
                                          _ptr(ub), _ptr(status)), "plp_bbox_batch_dev")
        return dict(lb=lb, ub=ub, status=status)
    A = _np(A)
    if A.ndim != 3:
        raise ValueError("A must be [B, m_max, d]")
    B, m_max, d = A.shape
    b = _np(b).reshape(B, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(B)
    # (inf / nan in the inputs: ValueError from the library, which checks them while staging -- plp_ctx_set_check_finite)
    lb = np.empty((B, d))
    ub = np.empty((B, d))
    status = np.empty(B, np.int32)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_bbox_batch(_lib.context().handle, B, m_max, d, _ptr(A), _ptr(b), _ptr(mm), _ptr(lb),
                                  _ptr(ub), _ptr(status)), "plp_bbox_batch")
    return dict(lb=lb, ub=ub, status=status)


def reduce_batch(A, b, m=None, abs_tol=1e-7, out=None):
    """Fused reduce() (polytope.py:1053-1163) of B non-minrep polytopes.

    -> dict(keep uint64[B] (bit i = input row i kept), flags int32[B] (RF_*), r[B], xc[B,d],
            nlp int32[B] = LPs the reference would have issued for that polytope)
    Polytopes of more than 64 rows (m_max > 64; the reference has no row limit): keep is [B, W], W = ceil(m_max / 64)
    words per polytope (plp_reduce_wide_batch; `keep_to_bool` reads either shape).
    `out` (device path only): a dict of preallocated, contiguous CUDA tensors keep int64[B], flags
    int32[B], r float64[B], xc float64[B,d], nlp int32[B] to write into (e.g. views of one exchange buffer,
    polytope_amd.dist.ResultBuffer).
    """
    lib = _lib.load()
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        B, m_max, d = A.shape
        if m_max > 64:
            if out is not None:
                raise ValueError("reduce_batch: `out` is for polytopes of up to 64 rows (one keep word each)")
            W = (m_max + 63) // 64
            keep = torch.empty((B, W), dtype=torch.int64, device=A.device)
            flags = torch.empty((B,), dtype=torch.int32, device=A.device)
            r = torch.empty((B,), dtype=torch.float64, device=A.device)
            xc = torch.empty((B, d), dtype=torch.float64, device=A.device)
            nlp = torch.empty((B,), dtype=torch.int32, device=A.device)
            _lib.check(lib.plp_reduce_wide_batch_dev(ctx.handle, stream, B, m_max, d, _ptr(A), _ptr(b), _ptr(m),
                                                     float(abs_tol), _ptr(keep), _ptr(flags), _ptr(r), _ptr(xc), _ptr(nlp)),
                       "plp_reduce_wide_batch_dev")
            return dict(keep=keep, flags=flags, r=r, xc=xc, nlp=nlp)
        if out is not None:
            keep, flags, r, xc, nlp = out["keep"], out["flags"], out["r"], out["xc"], out["nlp"]
            want = ((keep, torch.int64, (B,)), (flags, torch.int32, (B,)), (r, torch.float64, (B,)),
                    (xc, torch.float64, (B, d)), (nlp, torch.int32, (B,)))
            for t, dt, shp in want:
                if t.dtype != dt or tuple(t.shape) != shp or not t.is_contiguous() or t.device != A.device:
                    raise ValueError("reduce_batch: `out` tensors must be contiguous CUDA tensors of the documented "
                                     "dtype and shape")
        else:
            keep = torch.empty((B,), dtype=torch.int64, device=A.device)
            flags = torch.empty((B,), dtype=torch.int32, device=A.device)
            r = torch.empty((B,), dtype=torch.float64, device=A.device)
            xc = torch.empty((B, d), dtype=torch.float64, device=A.device)
            nlp = torch.empty((B,), dtype=torch.int32, device=A.device)
        _lib.check(lib.plp_reduce_batch_dev(ctx.handle, stream, B, m_max, d, _ptr(A), _ptr(b), _ptr(m), float(abs_tol),
                                            _ptr(keep), _ptr(flags), _ptr(r), _ptr(xc), _ptr(nlp)),
                   "plp_reduce_batch_dev")
        return dict(keep=keep, flags=flags, r=r, xc=xc, nlp=nlp)
    A = _np(A)
    if A.ndim != 3:
        raise ValueError("A must be [B, m_max, d]")
    B, m_max, d = A.shape
    b = _np(b).reshape(B, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(B)
    # (inf / nan in the inputs: ValueError from the library, which checks them while staging -- plp_ctx_set_check_finite)
    wide = m_max > 64
    keep = np.empty((B, (m_max + 63) // 64) if wide else B, np.uint64)
    flags = np.empty(B, np.int32)
    r = np.empty(B)
    xc = np.empty((B, d))
    nlp = np.empty(B, np.int32)
    fn, name = (lib.plp_reduce_wide_batch, "plp_reduce_wide_batch") if wide else (lib.plp_reduce_batch, "plp_reduce_batch")
    _count_h2d(A, b, mm)
    _lib.check(fn(_lib.context().handle, B, m_max, d, _ptr(A), _ptr(b), _ptr(mm), float(abs_tol),
                  _ptr(keep), _ptr(flags), _ptr(r), _ptr(xc), _ptr(nlp)), name)
    return dict(keep=keep, flags=flags, r=r, xc=xc, nlp=nlp)


def reduce_simplex_runs(device=None, reset=False, stream=None):
    """Number of LPs that ran the simplex in the fused reduce launches of this process's context for `device` since the
    counter was last reset (include/plp.h: plp_reduce_counters) -- `nlp` counts the LPs the reference issues, of which
    the presolve settles a part without a simplex run.  The first call switches the counting on and returns 0.
    `stream`: a torch stream (default: torch's current stream when torch is loaded with a GPU, else the HIP default)."""
    lib = _lib.load()
    ctx = _lib.context(device)
    if stream is None:
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(ctx.device)
    sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
    out = C.c_uint64(0)
    _lib.check(lib.plp_reduce_counters(ctx.handle, sp, C.cast(C.byref(out), C.c_void_p), 1 if reset else 0),
               "plp_reduce_counters")
    return int(out.value)


def verify_careful_lps(device=None, stream=None):
    """How many LPs of the LAST lpsolve_batch / cheby_ball_batch / bbox_batch call did not pass the verifier's certificate (or
    were reported unbounded / at a limit) and were solved again by the careful double-double engine (include/plp.h:
    plp_verify_counters).  `stream`: a torch stream (default: torch's current stream when torch is loaded with a GPU; calls
    with numpy arrays run on the context's own stream, which the library falls back to).  Blocks until that batch is done."""
    lib = _lib.load()
    ctx = _lib.context(device)
    if stream is None:
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(ctx.device)
    sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
    out = C.c_int64(0)
    _lib.check(lib.plp_verify_counters(ctx.handle, sp, C.cast(C.byref(out), C.c_void_p)), "plp_verify_counters")
    return int(out.value)


def lp_histograms(result):
    """Status and pivot-count histograms of an lpsolve_batch result (counters): dict(status={code: count},
    iters=(edges, counts) over the engines' pivot counts -- LPs the careful engine re-solved keep the engine's count)."""
    st = result["status"]
    it = result.get("iters")
    st = st.cpu().numpy() if hasattr(st, "cpu") else np.asarray(st)
    codes, cnt = np.unique(st, return_counts=True)
    out = {"status": {int(c): int(n) for c, n in zip(codes, cnt)}}
    if it is not None:
        it = it.cpu().numpy() if hasattr(it, "cpu") else np.asarray(it)
        edges = np.array([0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 1 << 30])
        out["iters"] = (edges.tolist(), np.histogram(it, bins=edges)[0].tolist())
    return out


def keep_to_bool(keep, m_max):
    """uint64 keep masks (one word per polytope, or [B, W] words for more than 64 rows) -> bool[B, m_max]."""
    keep = np.asarray(keep).astype(np.uint64)
    if keep.ndim == 1:
        keep = keep[:, None]
    rows = np.arange(m_max)
    words = keep[:, rows // 64]
    return ((words >> (rows % 64).astype(np.uint64)[None, :]) & np.uint64(1)).astype(bool)


def contains_batch(A, b, X, abs_tol=1e-7, m=None, region=True):
    """Containment of the N column vectors X[d, N] in P polytopes (polytope.py:206-218, :732-746).

    region=True  -> uint8[N]    OR over the polytopes (Region.contains; all P*N tests evaluated)
    region=False -> uint8[P, N] one row per polytope (Polytope.contains)
    """
    lib = _lib.load()
    mode = 0 if region else 1
    if _is_torch(X):
        torch, ctx, stream = _torch_stream_ctx(X)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        X = _tprep(torch, X, torch.float64)
        m = _tprep(torch, m, torch.int32)
        P, m_max, d = A.shape
        if X.shape[0] != d:
            raise ValueError("points should be column vectors")
        N = X.shape[1]
        out = torch.empty((N,) if region else (P, N), dtype=torch.uint8, device=X.device)
        _lib.check(lib.plp_contains_dev(ctx.handle, stream, P, m_max, d, _ptr(A), _ptr(b), _ptr(m), N, _ptr(X),
                                        float(abs_tol), mode, _ptr(out)), "plp_contains_dev")
        return out
    A = _np(A)
    P, m_max, d = A.shape
    b = _np(b).reshape(P, m_max)
    X = _np(X)
    if X.ndim != 2 or X.shape[0] != d:
        raise ValueError("points should be column vectors")
    N = X.shape[1]
    mm = None if m is None else _np(m, np.int32).reshape(P)
    out = np.zeros((N,) if region else (P, N), np.uint8)
    _count_h2d(A, b, mm, X)
    _lib.check(lib.plp_contains(_lib.context().handle, P, m_max, d, _ptr(A), _ptr(b), _ptr(mm), N, _ptr(X),
                                float(abs_tol), mode, _ptr(out)), "plp_contains")
    return out


def assign_batch(X, normals, offsets, abs_tol=1e-7):
    """quickhull outside-set assignment + furthest point (quickhull.py:87-102,117-121,224-245).

    X[N, d] points (rows), normals[F, d], offsets[F]
    -> dict(facet int32[N] (-1 inside), dist[N], argmax int64[F] (-1 none), maxd[F])
    """
    lib = _lib.load()
    if _is_torch(X):
        torch, ctx, stream = _torch_stream_ctx(X)
        X = _tprep(torch, X, torch.float64)
        normals = _tprep(torch, normals, torch.float64)
        offsets = _tprep(torch, offsets, torch.float64)
        N, d = X.shape
        F = normals.shape[0]
        fop = torch.empty((N,), dtype=torch.int32, device=X.device)
        dist = torch.empty((N,), dtype=torch.float64, device=X.device)
        am = torch.empty((F,), dtype=torch.int64, device=X.device)
        mx = torch.empty((F,), dtype=torch.float64, device=X.device)
        _lib.check(lib.plp_assign_dev(ctx.handle, stream, N, d, _ptr(X), F, _ptr(normals), _ptr(offsets),
                                      float(abs_tol), _ptr(fop), _ptr(dist), _ptr(am), _ptr(mx)), "plp_assign_dev")
        return dict(facet=fop, dist=dist, argmax=am, maxd=mx)
    X = _np(X)
    N, d = X.shape
    normals = _np(normals).reshape(-1, d)
    F = normals.shape[0]
    offsets = _np(offsets).reshape(F)
    fop = np.empty(N, np.int32)
    dist = np.empty(N)
    am = np.empty(F, np.int64)
    mx = np.empty(F)
    _count_h2d(X, normals, offsets)
    _lib.check(lib.plp_assign(_lib.context().handle, N, d, _ptr(X), F, _ptr(normals), _ptr(offsets), float(abs_tol),
                              _ptr(fop), _ptr(dist), _ptr(am), _ptr(mx)), "plp_assign")
    return dict(facet=fop, dist=dist, argmax=am, maxd=mx)


class HullSession:
    """Outside sets of one quickhull run, resident on the device (include/plp.h: plp_hull_*).

    The N points are uploaded once; per iteration `reassign` moves the points owned by the dead
    (visible) facets to the new facets and returns, per new facet, how many points it received and
    which one is furthest (quickhull.py:273-283, :311-336, :87-102).
    """

    def __init__(self, X):
        lib = _lib.load()
        X = _np(X)
        if X.ndim != 2:
            raise ValueError("points must be an (N, d) array")
        _finite_or_raise("HullSession", X)
        self.N, self.d = X.shape
        self._ctx = _lib.context()
        h = C.c_void_p()
        _lib.check(lib.plp_hull_create(self._ctx.handle, self.N, self.d, _ptr(X), C.byref(h)), "plp_hull_create")
        self._h = h

    def drop(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64).ravel()
        _lib.check(_lib.load().plp_hull_drop(self._h, idx.size, _ptr(idx)), "plp_hull_drop")

    def reassign(self, dead_ids, normals, offsets, abs_tol=1e-7):
        """-> (new_id0, count int64[n_new], argmax int64[n_new] (-1 none), maxd[n_new])"""
        dead = np.ascontiguousarray(dead_ids, dtype=np.int32).ravel()
        normals = _np(normals).reshape(-1, self.d)
        n_new = normals.shape[0]
        offsets = _np(offsets).reshape(n_new)
        am = np.empty(n_new, np.int64)
        mx = np.empty(n_new)
        cnt = np.empty(n_new, np.int64)
        id0 = C.c_int32(0)
        _lib.check(_lib.load().plp_hull_reassign(self._h, dead.size, _ptr(dead), n_new, _ptr(normals), _ptr(offsets),
                                                 float(abs_tol), C.byref(id0), _ptr(am), _ptr(mx), _ptr(cnt)),
                   "plp_hull_reassign")
        return int(id0.value), cnt, am, mx

    def read(self):
        """-> (owner int32[N], dist[N]) copied to the host"""
        owner = np.empty(self.N, np.int32)
        dist = np.empty(self.N)
        _lib.check(_lib.load().plp_hull_read(self._h, _ptr(owner), _ptr(dist)), "plp_hull_read")
        return owner, dist

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().plp_hull_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def hull_reassign_dev(X, owner, dist, dead, new_id0, normals, offsets, abs_tol=1e-7):
    """Stateless form on torch CUDA tensors (updates owner/dist in place) -> dict(count, argmax, maxd)."""
    lib = _lib.load()
    torch, ctx, stream = _torch_stream_ctx(X)
    N, d = X.shape
    n_new = normals.shape[0]
    am = torch.empty((n_new,), dtype=torch.int64, device=X.device)
    mx = torch.empty((n_new,), dtype=torch.float64, device=X.device)
    cnt = torch.empty((n_new,), dtype=torch.int64, device=X.device)
    _lib.check(lib.plp_hull_reassign_dev(ctx.handle, stream, N, d, _ptr(X), _ptr(owner), _ptr(dist), _ptr(dead),
                                         int(new_id0), n_new, _ptr(normals), _ptr(offsets), float(abs_tol),
                                         _ptr(am), _ptr(mx), _ptr(cnt)), "plp_hull_reassign_dev")
    return dict(count=cnt, argmax=am, maxd=mx)


def adjacent_pairs(A, b, m=None, abs_tol=1e-7):
    """Adjacency matrix of n single-polytope cells (prop2partition.py:46-63 over polytope.py:1843-1866):
    uint8[n, n], symmetric, ones on the diagonal.  A[n, m_max, d], b[n, m_max]; 2*m_max <= 64, d <= 16.
    The n(n-1)/2 stacked, abs_tol-inflated pair LPs are formed on the device."""
    lib = _lib.load()
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        n, m_max, d = A.shape
        adj = torch.empty((n, n), dtype=torch.uint8, device=A.device)
        _lib.check(lib.plp_adjacent_pairs_dev(ctx.handle, stream, n, m_max, d, _ptr(A), _ptr(b), _ptr(m),
                                              float(abs_tol), _ptr(adj)), "plp_adjacent_pairs_dev")
        return adj
    A = _np(A)
    n, m_max, d = A.shape
    b = _np(b).reshape(n, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(n)
    _finite_or_raise("adjacent_pairs", A, b)
    adj = np.zeros((n, n), np.uint8)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_adjacent_pairs(_lib.context().handle, n, m_max, d, _ptr(A), _ptr(b), _ptr(mm), float(abs_tol),
                                      _ptr(adj)), "plp_adjacent_pairs")
    return adj


def overlap_pairs(A, b, m=None, abs_tol=1e-7):
    """uint8[n, n]: 1 where the intersection of cells i and j is full-dimensional (Chebyshev radius of the
    stacked rows > abs_tol) -- the pair test of Partition.are_disjoint (prop2partition.py:146-149); ones
    on the diagonal.  A[n, m_max, d], b[n, m_max]; 2*m_max <= 64, d <= 16."""
    lib = _lib.load()
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        n, m_max, d = A.shape
        out = torch.empty((n, n), dtype=torch.uint8, device=A.device)
        _lib.check(lib.plp_overlap_pairs_dev(ctx.handle, stream, n, m_max, d, _ptr(A), _ptr(b), _ptr(m),
                                             float(abs_tol), _ptr(out)), "plp_overlap_pairs_dev")
        return out
    A = _np(A)
    n, m_max, d = A.shape
    b = _np(b).reshape(n, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(n)
    _finite_or_raise("overlap_pairs", A, b)
    out = np.zeros((n, n), np.uint8)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_overlap_pairs(_lib.context().handle, n, m_max, d, _ptr(A), _ptr(b), _ptr(mm), float(abs_tol),
                                     _ptr(out)), "plp_overlap_pairs")
    return out


def overlap_cross(A, b, n1, m=None, thresh=1e-7):
    """uint8[n1, n - n1]: 1 where the stack [cell a; cell c] of cell a < n1 of the table and cell c >= n1 has a Chebyshev
    radius > thresh -- the opening scan of region_diff (polytope.py:2148-2158) for every (minuend member, subtrahend
    cell) pair at once.  A[n, m_max, d], b[n, m_max]; 2 * m_max <= 64, d <= 16."""
    lib = _lib.load()
    n1 = int(n1)
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        n, m_max, d = A.shape
        out = torch.empty((n1, n - n1), dtype=torch.uint8, device=A.device)
        _lib.check(lib.plp_overlap_cross_dev(ctx.handle, stream, n1, n - n1, m_max, d, _ptr(A), _ptr(b), _ptr(m),
                                             float(thresh), _ptr(out)), "plp_overlap_cross_dev")
        return out
    A = _np(A)
    n, m_max, d = A.shape
    b = _np(b).reshape(n, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(n)
    _finite_or_raise("overlap_cross", A, b)
    out = np.zeros((n1, n - n1), np.uint8)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_overlap_cross(_lib.context().handle, n1, n - n1, m_max, d, _ptr(A), _ptr(b), _ptr(mm), float(thresh),
                                     _ptr(out)), "plp_overlap_cross")
    return out


def adjacent_pairs_range(A, b, pair_lo, pair_hi, m=None, abs_tol=1e-7):
    """Adjacency of the cell pairs pair_lo <= p < pair_hi (p = i (i - 1) / 2 + j, j < i) -> uint8[pair_hi - pair_lo];
    one rank's shard of the O(n^2) loop of find_adjacent_regions (prop2partition.py:57-61)."""
    lib = _lib.load()
    lo, hi = int(pair_lo), int(pair_hi)
    if _is_torch(A):
        torch, ctx, stream = _torch_stream_ctx(A)
        A = _tprep(torch, A, torch.float64)
        b = _tprep(torch, b, torch.float64)
        m = _tprep(torch, m, torch.int32)
        n, m_max, d = A.shape
        out = torch.empty((max(hi - lo, 0),), dtype=torch.uint8, device=A.device)
        _lib.check(lib.plp_adjacent_pairs_range_dev(ctx.handle, stream, n, m_max, d, _ptr(A), _ptr(b), _ptr(m),
                                                    float(abs_tol), lo, hi, _ptr(out)), "plp_adjacent_pairs_range_dev")
        return out
    A = _np(A)
    n, m_max, d = A.shape
    b = _np(b).reshape(n, m_max)
    mm = None if m is None else _np(m, np.int32).reshape(n)
    _finite_or_raise("adjacent_pairs_range", A, b)
    out = np.zeros(max(hi - lo, 0), np.uint8)
    _count_h2d(A, b, mm)
    _lib.check(lib.plp_adjacent_pairs_range(_lib.context().handle, n, m_max, d, _ptr(A), _ptr(b), _ptr(mm),
                                            float(abs_tol), lo, hi, _ptr(out)), "plp_adjacent_pairs_range")
    return out


def selftest(group_size):
    """Cross-lane primitive self-test -> (out_d[128], out_u[128]) (see plp_points.hip)."""
    lib = _lib.load()
    od = np.empty(128)
    ou = np.empty(128, np.uint32)
    _lib.check(lib.plp_selftest(_lib.context().handle, int(group_size), _ptr(od), _ptr(ou)), "plp_selftest")
    return od, ou


def region_diff_search(A, b, m, mi, abs_tol=1e-7):
    """The search of region_diff (polytope.py:2201-2281) run by the library on the table A[m + 2M, d], b[m + 2M]
    (poly's m rows, the cells' new rows, their negations; rows already unit length) with mi[j] new rows per cell.

    -> (pieces, stats): pieces = list of (kind, rows) in the reference's order, kind 0 = Polytope(A[rows], b[rows]) as
    is, 1 = reduce() of it; stats = dict(lps, batches).  One launch + one synchronisation per visited node; the LPs
    are gathered on the device from the resident table by row index (include/plp.h: plp_region_diff_search)."""
    lib = _lib.load()
    A = _np(A)
    b = _np(b).ravel()
    mi = _np(mi, np.int32).ravel()
    nrows, d = A.shape
    if nrows != m + 2 * int(mi.sum()) or b.size != nrows:
        raise ValueError("region_diff_search: table must have m + 2 * sum(mi) rows")
    _finite_or_raise("region_diff_search", A, b)
    h = C.c_void_p()
    _count_h2d(A, b)
    _lib.check(lib.plp_region_diff_search(_lib.context().handle, d, int(m), int(mi.size), _ptr(mi), _ptr(A), _ptr(b),
                                          float(abs_tol), C.byref(h)), "plp_region_diff_search")
    try:
        nl, nr, nlp, nb = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.plp_rdiff_result_sizes(h, C.byref(nl), C.byref(nr), C.byref(nlp), C.byref(nb)), "plp_rdiff_result_sizes")
        kind = np.empty(max(nl.value, 1), np.int32)
        off = np.empty(nl.value + 1, np.int32)
        rows = np.empty(max(nr.value, 1), np.int32)
        _lib.check(lib.plp_rdiff_result_copy(h, _ptr(kind), _ptr(off), _ptr(rows)), "plp_rdiff_result_copy")
    finally:
        lib.plp_rdiff_result_free(h)
    pieces = [(int(kind[k]), rows[off[k]:off[k + 1]].copy()) for k in range(nl.value)]
    return pieces, dict(lps=int(nlp.value), batches=int(nb.value))


_DGESV = None


def _lapack_dgesv_pointer():
    """Address of the dgesv numpy.linalg.solve's results agree with bit for bit (scipy's LAPACK; checked in the
    tests), or None when scipy is not importable -- the library then uses its own LU."""
    global _DGESV
    if _DGESV is None:
        try:
            from scipy.linalg import cython_lapack
            cap = cython_lapack.__pyx_capi__["dgesv"]
            C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
            C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
            C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
            C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
            _DGESV = C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap)) or 0
        except Exception:
            _DGESV = 0
    return _DGESV or None


def quickhull_run(X0, simplex, abs_tol=1e-7):
    """Quickhull's main loop in the library (include/plp.h: plp_quickhull_run): X0[N, d] translated points, simplex =
    indices of the start simplex -> (normals[n, d], offsets[n], verts int64[n, d], stats) of the hull's facets in the
    reference's order."""
    lib = _lib.load()
    X0 = _np(X0)
    N, d = X0.shape
    simplex = np.ascontiguousarray(simplex, dtype=np.int64).ravel()
    if simplex.size != d + 1:
        raise ValueError("quickhull_run: the start simplex has d + 1 points")
    _finite_or_raise("quickhull_run", X0)
    h = C.c_void_p()
    rc = lib.plp_quickhull_run(_lib.context().handle, N, d, _ptr(X0), _ptr(simplex), float(abs_tol),
                               C.c_void_p(_lapack_dgesv_pointer()), C.byref(h))
    if rc:
        msg = lib.plp_quickhull_last_error().decode("utf-8", "replace")
        if "Singular matrix" in msg:
            raise np.linalg.LinAlgError("Singular matrix")
        if "identical vertices" in msg:
            raise RuntimeError(msg)
        _lib.check(rc, "plp_quickhull_run")
    try:
        nf, it, made = C.c_int64(), C.c_int64(), C.c_int64()
        lib.plp_qh_result_sizes(h, C.byref(nf), C.byref(it), C.byref(made))
        normals = np.empty((nf.value, d))
        offsets = np.empty(nf.value)
        verts = np.empty((nf.value, d), np.int64)
        lib.plp_qh_result_copy(h, _ptr(normals), _ptr(offsets), _ptr(verts))
    finally:
        lib.plp_qh_result_free(h)
    return normals, offsets, verts, dict(iterations=int(it.value), facets_made=int(made.value))
