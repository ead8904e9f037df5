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
from collections import OrderedDict, deque

import numpy as np

from . import solvers

logger = logging.getLogger(__name__)


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


class _Facet:
    __slots__ = ("fid", "verts", "normal", "offset", "neighbors", "count", "far")

    def __init__(self, verts, normal, offset):
        self.fid = -1            # device-side facet id (owner value of the points outside it)
        self.verts = verts       # d point indices
        self.normal = normal     # unit outward normal (d,)
        self.offset = offset     # n.x = offset on the facet (translated coordinates)
        self.neighbors = []
        self.count = 0           # points currently outside this facet
        self.far = -1            # index of the furthest of them


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


def _rank(M, tol):
    return int(np.sum(np.linalg.svd(M, compute_uv=False) > tol))


def quickhull(POINTS, abs_tol=1e-7, session_factory=None):
    """Compute the convex hull of a set of points.

    @param POINTS: a n*d np array where each row denotes a point
    @param session_factory: (not in the reference) callable X0 -> outside-set session; default: the
        device-resident HullSession of the 'hip' backend.  polytope_amd.dist.quickhull_sharded passes
        the multi-GPU session here.

    @return: A,b,vertices: `A` and `b` describing the convex hull polytope as A x <= b
        (H-representation). `vertices` is an array of all the points in the convex hull
        (V-representation), or None with empty `A`, `b` if the hull is not fully dimensional.
    """
    POINTS = np.asarray(POINTS).astype("float")
    if POINTS.ndim != 2:
        raise ValueError("quickhull: POINTS must be an (n, d) array")
    npt, dim = POINTS.shape
    if npt <= dim:
        return np.array([]), np.array([]), None  # convex hull is empty
    # full-dimensional?  (the singular values of the reference's check :157-163, thin SVD)
    if _rank((POINTS - POINTS[0, :]).T, 1e-15) < dim:
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

    def make_facets(vert_lists):
        idx = np.asarray(vert_lists, dtype=np.int64).reshape(-1, dim)
        n, off = _hyperplanes(X0[idx])
        return [_Facet(list(map(int, idx[k])), n[k], float(off[k])) for k in range(idx.shape[0])]

    order = list(range(dim + 1))
    first = make_facets([[ind[j] for j in order if j != i] for i in range(dim + 1)])
    facets = OrderedDict()  # the reference's Forg, insertion ordered; key = id(facet object)
    for f in first:
        facets[id(f)] = f

    def result():
        flist = list(facets.values())
        A = np.array([f.normal for f in flist]).reshape(len(flist), dim)
        b = np.array([f.offset for f in flist])
        vid = np.unique(np.concatenate([np.asarray(f.verts, dtype=np.int64) for f in flist]))
        vert = POINTS[vid]
        vert = vert[np.lexsort(vert.T[::-1])]  # np.unique row order of the reference (:352-354)
        return A, b + np.dot(A, xc), vert

    if npt == dim + 1:
        return result()
    for ii in range(dim + 1):  # in the starting simplex all facets are neighbours (:215-222)
        for jj in range(ii + 1, dim + 1):
            first[ii].neighbors.append(first[jj])
            first[jj].neighbors.append(first[ii])

    session = (session_factory or _open_session)(X0)
    try:
        pending = OrderedDict()  # the reference's F: facets with outside points, FIFO

        def hand_out(dead_ids, new):
            id0, count, argmax, _ = session.reassign(dead_ids, np.array([f.normal for f in new]),
                                                     np.array([f.offset for f in new]), abs_tol)
            for k, f in enumerate(new):
                f.fid = id0 + k
                f.count = int(count[k])
                f.far = int(argmax[k])
                if f.count > 0:
                    pending[id(f)] = f

        session.drop(ind)      # the simplex' own points are not candidates (:186)
        hand_out([0], first)   # facet id 0 owns every point initially
        while pending:
            facet = next(iter(pending.values()))
            p = facet.far
            session.drop([p])  # get_furthest() removes it from the facet's outside set (:87-102)
            facet.count -= 1
            xp = X0[p]
            # ---- visible set: breadth-first over neighbours with distance > abs_tol (:254-270)
            visible = [facet]
            in_visible = {id(facet)}
            seen = {id(facet)}
            queue = deque(facet.neighbors)
            queued = {id(f) for f in facet.neighbors}
            while queue:
                nb = queue.popleft()
                queued.discard(id(nb))
                seen.add(id(nb))
                if np.sum(nb.normal * xp) - nb.offset > abs_tol:
                    visible.append(nb)
                    in_visible.add(id(nb))
                    for nn in nb.neighbors:
                        if id(nn) not in seen and id(nn) not in queued:
                            queue.append(nn)
                            queued.add(id(nn))
            # ---- horizon: one new facet per (visible facet, non-visible neighbour) (:284-304)
            new_verts, outer = [], []
            for f1 in visible:
                for f2 in f1.neighbors:
                    if id(f2) in in_visible:
                        continue
                    other = set(f2.verts)
                    ridge = None
                    for ii in range(dim):
                        if f1.verts[ii] not in other:
                            ridge = [v for jj, v in enumerate(f1.verts) if jj != ii]
                            break
                    if ridge is None:  # same vertex set twice: degenerate input
                        raise RuntimeError("quickhull: neighbouring facets with identical vertices")
                    new_verts.append([p] + ridge)
                    outer.append(f2)
            new = make_facets(new_verts)
            for f, f2 in zip(new, outer):
                f.neighbors.append(f2)
                f2.neighbors.append(f)
            # ---- links among the new facets: two of them share p and d-2 ridge vertices (:305-310)
            by_subridge = {}
            for k, f in enumerate(new):
                ridge = f.verts[1:]
                for omit in range(len(ridge)):
                    key = frozenset(ridge[:omit] + ridge[omit + 1:])
                    by_subridge.setdefault(key, []).append(k)
            links = [set() for _ in new]
            for group in by_subridge.values():
                for a in group:
                    for c in group:
                        if a != c:
                            links[a].add(c)
            for k, f in enumerate(new):
                for c in sorted(links[k]):
                    f.neighbors.append(new[c])
            # ---- hand the pooled points to the new facets, retire the visible ones (:311-344)
            pooled = sum(f.count for f in visible)
            for f in new:
                facets[id(f)] = f
            if pooled > 0:
                hand_out([f.fid for f in visible if f.count > 0], new)  # facets without points own nothing
            for f1 in visible:
                for f2 in f1.neighbors:
                    f2.neighbors.remove(f1)
                pending.pop(id(f1), None)
                del facets[id(f1)]
                f1.neighbors = []
    finally:
        session.close()
    return result()
