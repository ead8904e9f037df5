"""Convex hull of a point set by Quickhull -- mirror of polytope/quickhull.py (`quickhull`,
reference :141-359) with the point work on the MI355X.

Split of labour
    device  the N points, resident for the whole run ([N][d] rows, translated so that the start
            simplex' centroid is the origin, reference :188-192), one owner facet and one distance
            per point.  Every pass over the points is one `plp_hull_reassign` call
            (include/plp.h): the initial assignment (:224-245), each iteration's pooling of the
            visible facets' outside points and their re-assignment to the new facets (:273-283,
            :311-336), and the furthest point of every new facet (:87-102).
    host    the facet graph (a few hundred facets): start simplex (:165-185), visibility search from
            the chosen point (:254-270), horizon ridges and new facets (:284-304), neighbour links
            (:305-310), retiring the visible facets (:337-344).  Facet hyperplanes come from the
            same (d+1)x(d+1) linear system as the reference's Facet (:61-85), solved for all new
            facets of an iteration in one batched LAPACK call.

Facet and neighbour ORDER follow the reference (first facet with outside points is processed
first; visibility is searched breadth-first over the neighbour lists; new facets are created in
(visible facet, neighbour) order), and the start simplex consumes numpy's global RNG exactly as
the reference does, so that with the same `np.random.seed` both produce the same rows in the same
order.  Neighbour detection works on point indices instead of coordinate comparison (:124-138):
identical for inputs without near-duplicate points.

With `solvers.default_solver != 'hip'` the point passes run in numpy with the reference's
arithmetic (this is the explicit CPU backend used by the CPU-only tests; nothing falls back to it).
"""
import logging
from collections import deque

import numpy as np

from . import solvers

logger = logging.getLogger(__name__)
_NATIVE_LOOP = True   # 'hip' backend: main loop in the library (False: the Python facet graph below, for A/B runs)


class _NumpySession:
    """Outside sets in numpy -- the reference's arithmetic (np.sum(n*p) - d, quickhull.py:117-121);
    interface of polytope_amd.batch.HullSession."""

    def __init__(self, X):
        self.X = np.ascontiguousarray(X, dtype=float)
        self.N, self.d = self.X.shape
        self.owner = np.zeros(self.N, np.int32)
        self.dist = np.zeros(self.N)
        self.next_id = 1

    def drop(self, idx):
        self.owner[np.asarray(idx, dtype=np.int64)] = -1

    def reassign(self, dead_ids, normals, offsets, abs_tol=1e-7):
        normals = np.asarray(normals, dtype=float).reshape(-1, self.d)
        offsets = np.asarray(offsets, dtype=float).ravel()
        n_new = normals.shape[0]
        id0 = self.next_id
        dead_ids = np.asarray(dead_ids, dtype=np.int32).ravel()
        if np.any(dead_ids < 0) or np.any(dead_ids >= id0):
            raise ValueError("dead facet id was never handed out")
        self.next_id += n_new
        pooled = np.nonzero(np.isin(self.owner, dead_ids))[0]
        count = np.zeros(n_new, np.int64)
        argmax = np.full(n_new, -1, np.int64)
        maxd = np.zeros(n_new)
        if pooled.size == 0:
            return id0, count, argmax, maxd
        P = self.X[pooled]
        D = np.empty((pooled.size, n_new))
        for f in range(n_new):  # sum over k in index order, as np.sum(n * p) does for d < 8
            D[:, f] = np.sum(normals[f][None, :] * P, axis=1) - offsets[f]
        out = D > abs_tol
        first = np.where(out.any(axis=1), out.argmax(axis=1), -1)
        self.owner[pooled] = np.where(first >= 0, id0 + first, -1).astype(np.int32)
        dd = np.where(first >= 0, D[np.arange(pooled.size), np.maximum(first, 0)], 0.0)
        self.dist[pooled] = dd
        for f in range(n_new):
            sel = np.nonzero(first == f)[0]
            count[f] = sel.size
            if sel.size:
                k = sel[np.argmax(dd[sel])]  # first maximum = lowest point index (pooled is ascending)
                argmax[f] = pooled[k]
                maxd[f] = dd[k]
        return id0, count, argmax, maxd

    def read(self):
        return self.owner.copy(), self.dist.copy()

    def close(self):
        pass


def _open_session(X):
    if solvers.default_solver == "hip":
        from .batch import HullSession
        return HullSession(X)
    return _NumpySession(X)


def _hyperplanes(V):
    """Unit outward normals and offsets of the facets with vertex coordinates V[k] (d x d each),
    from the linear system of the reference's Facet.__init__ (quickhull.py:66-85)."""
    k, d, _ = V.shape
    M = np.zeros((k, d + 1, d + 1))
    M[:, :d, :d] = V
    M[:, :d, d] = 1.0
    M[:, d, d] = -1.0
    rhs = np.zeros((k, d + 1, 1))
    rhs[:, d, 0] = 1.0
    sol = np.linalg.solve(M, rhs)[:, :, 0]
    xx = sol[:, :d]
    mult = np.sqrt(np.sum(xx ** 2, axis=1))
    n = xx / mult[:, None]
    dd = sol[:, d] / mult
    flip = np.sum(n * V[:, 0, :], axis=1) < 0
    n[flip] = -n[flip]
    return n, -dd


# ---- the reference's module-level objects (quickhull.py:43-139), for code that imports them.  The hull itself does not
# go through them: it keeps index arrays on the device (HullSession) instead of a list of point objects per facet.
class Facet(object):
    """Facet of an n-dimensional polyhedron that contains the origin (ref :43-102): `vertices` (n x n, one per row),
    `normal` (unit, pointing out, n x 1), `distance` (of the facet's plane from the origin), `outside` (a list of
    Outside_point), `neighbors`."""

    def __init__(self, points):
        points = np.asarray(points, dtype=float)
        self.outside = []
        self.vertices = points
        self.neighbors = []
        n, off = _hyperplanes(points[None, :, :])
        self.normal = n[0][:, None]
        self.distance = np.array([off[0]])

    def get_furthest(self):
        """Remove and return the outside point furthest from the facet; the first one among equals (ref :87-102)."""
        best = 0
        for i in range(1, len(self.outside)):
            if self.outside[best].distance < self.outside[i].distance:
                best = i
        return self.outside.pop(best)


class Outside_point(object):
    """Coordinates of a point and its distance to the facet it is assigned to (ref :105-114)."""

    def __init__(self, coordinates, distance):
        self.distance = distance
        self.coordinates = coordinates


def distance(p, fac1):
    """Signed distance of point `p` from the plane of `fac1` (ref :117-121): np.sum(n * p) - d."""
    return np.sum(np.asarray(fac1.normal).flatten() * np.asarray(p).flatten()) - fac1.distance


def is_neighbor(fac1, fac2, abs_tol=1e-7):
    """True if the two facets share d - 1 vertices (ref :124-139): vertices of fac1 that have a twin in fac2."""
    v1, v2 = np.asarray(fac1.vertices), np.asarray(fac2.vertices)
    dim = v1.shape[1]
    twin = np.all(np.abs(v1[:dim, None, :] - v2[None, :dim, :]) < abs_tol, axis=2)
    return int(np.count_nonzero(twin.any(axis=1))) == dim - 1


def _rank(M, tol):
    return int(np.sum(np.linalg.svd(M, compute_uv=False) > tol))


def _full_rank_clearly(D):
    """True when the rows of D (n x d, n >= d) clearly span R^d by the reference's ABSOLUTE test (singular values above
    1e-15, :157-163): the eigenvalues of the d x d Gram matrix D'D are the squared singular values; accepted only when the
    smallest is above 1e-6 of the largest (the Gram matrix of a million rows is good to ~1e-10 of its largest eigenvalue,
    so the smallest is then known to a few digits) AND above 1e-24, i.e. the smallest singular value is above 1e-12 --
    three orders clear of the reference's threshold.  Tiny-scale point sets (coordinates around 1e-13 and below), anything
    less clear, or not finite, go to the SVD itself.  One pass over the points instead of the thin SVD's several
    (1M x 3: 3 ms against 20)."""
    G = D.T @ D
    if not np.all(np.isfinite(G)):
        return False
    lam = np.linalg.eigvalsh(G)
    return bool(lam[0] > 1e-6 * lam[-1] and lam[0] > 1e-24)


_BLAS_CTL = None


def _blas_single_thread():
    """Context manager: numpy's BLAS limited to one thread for the duration of one hull.  The limit is the BLAS
    library's own thread count, i.e. PROCESS-WIDE: BLAS calls made by other Python threads meanwhile also run
    single-threaded (tens of milliseconds per hull).
    The rank check (an SVD) and the start simplex (matrix-vector products) wake OpenBLAS's worker pool, whose threads
    then spin for tens of milliseconds; measured on the 256-core host of the GPU box they stall the native main loop
    that follows for 60-90 ms somewhere inside a 40 ms hull (N = 100 000, d = 5).  These calls are far too small to
    gain from threads.  No-op when threadpoolctl is not importable."""
    global _BLAS_CTL
    if _BLAS_CTL is None:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS_CTL = ThreadpoolController()
        except Exception:  # pragma: no cover
            _BLAS_CTL = False
    if _BLAS_CTL:
        return _BLAS_CTL.limit(limits=1, user_api="blas")
    import contextlib
    return contextlib.nullcontext()


def quickhull(POINTS, abs_tol=1e-7, session_factory=None):
    """Compute the convex hull of a set of points (see `_quickhull`; numpy's BLAS is limited to one thread, process-wide,
    meanwhile: `_blas_single_thread`)."""
    with _blas_single_thread():
        return _quickhull(POINTS, abs_tol, session_factory)


def _quickhull(POINTS, abs_tol=1e-7, session_factory=None):
    """Compute the convex hull of a set of points.

    @param POINTS: a n*d np array where each row denotes a point
    @param session_factory: (not in the reference) callable X0 -> outside-set session; default: the
        device-resident HullSession of the 'hip' backend.  polytope_amd.dist.quickhull_sharded passes
        the multi-GPU session here.

    @return: A,b,vertices: `A` and `b` describing the convex hull polytope as A x <= b
        (H-representation). `vertices` is an array of all the points in the convex hull
        (V-representation), or None with empty `A`, `b` if the hull is not fully dimensional.
    """
    POINTS = np.asarray(POINTS, dtype=float)   # (never written to below: no copy of a float array)
    if POINTS.ndim != 2:
        raise ValueError("quickhull: POINTS must be an (n, d) array")
    npt, dim = POINTS.shape
    if npt <= dim:
        return np.array([]), np.array([]), None  # convex hull is empty
    # full-dimensional?  (the singular values of the reference's check :157-163, thin SVD -- after a one-pass screen)
    D0 = POINTS - POINTS[0, :]
    if not _full_rank_clearly(D0) and _rank(D0.T, 1e-15) < dim:
        logger.warning("convex hull is not fully dimensional, returning empty polytope")
        return np.array([]), np.array([]), None
    # ---- start simplex: extreme points in random directions (:165-185), same RNG stream
    rank = 0
    while rank < dim:
        ind = []
        for _ in range(dim + 1):
            rand = np.random.rand(dim) - 0.5
            test = np.dot(POINTS, rand)
            test[ind] = np.inf  # lowest projection among the points not taken yet
            ind.append(int(np.argmin(test)))
        startsimplex = POINTS[ind, :]
        rank = _rank((startsimplex - startsimplex[0, :]).T, 1e-10)
    xc = np.zeros(dim)
    for ii in range(dim + 1):
        xc += startsimplex[ii, :] / (dim + 1)
    X0 = POINTS - xc  # all coordinates below are relative to the simplex centroid (:188-192)

    if npt > dim + 1 and session_factory is None and solvers.default_solver == "hip" and _NATIVE_LOOP:
        # the main loop runs in the library (csrc/plp_quickhull_host.hip): same order, same arithmetic, the facet graph
        # in native code instead of Python lists and dicts
        from .batch import quickhull_run
        FNl, FOl, Vl, _ = quickhull_run(X0, ind, abs_tol)
        vid = np.unique(Vl.ravel())
        vert = POINTS[vid]
        vert = vert[np.lexsort(vert.T[::-1])]
        return FNl, FOl + np.dot(FNl, xc), vert
    # ---- facet table, indexed by slot (a facet keeps its slot for life; slots are never reused)
    FN = np.empty((64, dim))   # unit outward normals
    FO = np.empty(64)          # offsets: n.x = offset on the facet (translated coordinates)
    verts, nbrs = [], []       # slot -> d point indices / neighbour slots in the reference's list order
    cnt, far, fid = [], [], [] # slot -> points outside it, the furthest of them, device-side facet id

    def add_facets(vert_lists):
        nonlocal FN, FO
        idx = np.asarray(vert_lists, dtype=np.int64).reshape(-1, dim)
        n, off = _hyperplanes(X0[idx])
        s0, k = len(verts), idx.shape[0]
        while s0 + k > FO.shape[0]:
            FN = np.concatenate([FN, np.empty_like(FN)])
            FO = np.concatenate([FO, np.empty_like(FO)])
        FN[s0:s0 + k] = n
        FO[s0:s0 + k] = off
        for row in idx.tolist():
            verts.append(row)
            nbrs.append([])
            cnt.append(0)
            far.append(-1)
            fid.append(-1)
        return s0, k

    s0, k0 = add_facets([[ind[j] for j in range(dim + 1) if j != i] for i in range(dim + 1)])
    facets = dict.fromkeys(range(s0, s0 + k0))  # the reference's Forg: insertion ordered set of live slots

    def result():
        slots = list(facets)
        A = FN[slots].reshape(len(slots), dim)
        b = FO[slots]
        vid = np.unique(np.concatenate([np.asarray(verts[f], dtype=np.int64) for f in slots]))
        vert = POINTS[vid]
        vert = vert[np.lexsort(vert.T[::-1])]  # np.unique row order of the reference (:352-354)
        return A, b + np.dot(A, xc), vert

    if npt == dim + 1:
        return result()
    for ii in range(dim + 1):  # in the starting simplex all facets are neighbours (:215-222)
        for jj in range(ii + 1, dim + 1):
            nbrs[ii].append(jj)
            nbrs[jj].append(ii)

    session = (session_factory or _open_session)(X0)
    try:
        pending = {}  # the reference's F: facets with outside points, FIFO (insertion ordered)

        def hand_out(dead_ids, s0, k):
            id0, count, argmax, _ = session.reassign(dead_ids, FN[s0:s0 + k], FO[s0:s0 + k], abs_tol)
            count, argmax = count.tolist(), argmax.tolist()
            for j in range(k):
                f = s0 + j
                fid[f] = id0 + j
                cnt[f] = count[j]
                far[f] = argmax[j]
                if count[j] > 0:
                    pending[f] = None

        session.drop(ind)           # the simplex' own points are not candidates (:186)
        hand_out([0], s0, k0)       # facet id 0 owns every point initially
        while pending:
            facet = next(iter(pending))
            p = far[facet]
            session.drop([p])  # get_furthest() removes it from the facet's outside set (:87-102)
            cnt[facet] -= 1
            # distance of p to every facet made so far, one pass with distance()'s arithmetic (:117-121)
            nf = len(verts)
            vis = (np.sum(FN[:nf] * X0[p], axis=1) - FO[:nf]) > abs_tol
            # ---- visible set: breadth-first over neighbours with distance > abs_tol (:254-270)
            visible = [facet]
            in_visible = {facet}
            seen = {facet}
            queue = deque(nbrs[facet])
            queued = set(nbrs[facet])
            while queue:
                nb = queue.popleft()
                queued.discard(nb)
                seen.add(nb)
                if vis[nb]:
                    visible.append(nb)
                    in_visible.add(nb)
                    for nn in nbrs[nb]:
                        if nn not in seen and nn not in queued:
                            queue.append(nn)
                            queued.add(nn)
            # ---- horizon: one new facet per (visible facet, non-visible neighbour) (:284-304)
            new_verts, outer = [], []
            for f1 in visible:
                v1 = verts[f1]
                for f2 in nbrs[f1]:
                    if f2 in in_visible:
                        continue
                    other = set(verts[f2])
                    ridge = None
                    for ii in range(dim):
                        if v1[ii] not in other:
                            ridge = v1[:ii] + v1[ii + 1:]
                            break
                    if ridge is None:  # same vertex set twice: degenerate input
                        raise RuntimeError("quickhull: neighbouring facets with identical vertices")
                    new_verts.append([p] + ridge)
                    outer.append(f2)
            s0, k = add_facets(new_verts)
            for j, f2 in enumerate(outer):
                nbrs[s0 + j].append(f2)
                nbrs[f2].append(s0 + j)
            # ---- links among the new facets: two of them share p and d-2 ridge vertices (:305-310)
            by_subridge = {}
            for j in range(k):
                ridge = verts[s0 + j][1:]
                for omit in range(dim - 1):
                    by_subridge.setdefault(frozenset(ridge[:omit] + ridge[omit + 1:]), []).append(j)
            links = [set() for _ in range(k)]
            for group in by_subridge.values():
                if len(group) > 1:
                    for a in group:
                        for c in group:
                            if a != c:
                                links[a].add(c)
            for j in range(k):
                nbrs[s0 + j].extend(s0 + c for c in sorted(links[j]))
            # ---- hand the pooled points to the new facets, retire the visible ones (:311-344)
            for j in range(k):
                facets[s0 + j] = None
            if sum(cnt[f] for f in visible) > 0:
                hand_out([fid[f] for f in visible if cnt[f] > 0], s0, k)  # facets without points own nothing
            for f1 in visible:
                for f2 in nbrs[f1]:
                    nbrs[f2].remove(f1)
                pending.pop(f1, None)
                del facets[f1]
                nbrs[f1] = []
    finally:
        session.close()
    return result()
