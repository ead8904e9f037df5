"""Adjacency of the regions of a partition -- mirror of the O(n^2) pair loop of the reference
(polytope/prop2partition.py:46-63, `find_adjacent_regions`), which calls `is_adjacent` on every
pair (i, j < i) and therefore issues one Chebyshev LP per pair of member polytopes
(polytope/polytope.py:1843-1866).  Here all pair LPs go to the device as ONE batch.

Also the other two O(n^2) pair loops of that module: `Partition.are_disjoint` (:123-192, one
intersect + is_fulldim per pair) and `MetricPartition.compute_adj` (:244-306, is_adjacent on every ordered
pair plus a comparison with the previous matrix), as functions over a list of regions.

Only these pair computations are rebuilt; the `Partition` / `MetricPartition` containers of the
reference are plain Python bookkeeping around them.
"""
import logging
import numpy as np
import scipy.sparse as sp

from . import polytope as pc
from . import solvers

logger = logging.getLogger(__name__)


def _regions_of(partition):
    return partition.regions if hasattr(partition, "regions") else list(partition)


def _uniform_single_cells(regions):
    """All regions are single polytopes (or 1-member Regions) with identical (m, d)?"""
    cells = []
    for r in regions:
        if isinstance(r, pc.Region):
            if len(r) != 1:
                return None
            r = r.list_poly[0]
        cells.append(r)
    shapes = {c.A.shape for c in cells}
    if len(shapes) != 1:
        return None
    m, d = next(iter(shapes))
    if 2 * m > 64 or d > 16 or m < 1:
        return None
    return cells


def adjacency_matrix_dense(regions, abs_tol=pc.ABS_TOL):
    """n x n int8 adjacency (1 on the diagonal) -- is_adjacent(a, b, overlap=True) for every pair."""
    n = len(regions)
    adj = np.eye(n, dtype=np.int8)
    if n < 2:
        return adj
    ii, jj = np.tril_indices(n, -1)
    cells = _uniform_single_cells(regions) if solvers.default_solver == "hip" else None
    if cells is not None:
        # the n(n-1)/2 stacked, abs_tol-inflated pair LPs (polytope.py:1860-1866) are formed on the device
        from .batch import adjacent_pairs
        A = np.stack([c.A for c in cells])
        b = np.stack([c.b for c in cells])
        return adjacent_pairs(A, b, abs_tol=abs_tol).astype(np.int8)
    flags = pc.is_adjacent_pairs([(regions[i], regions[j]) for i, j in zip(ii, jj)], abs_tol=abs_tol)
    adj[ii[flags], jj[flags]] = 1
    adj[jj[flags], ii[flags]] = 1
    return adj


def overlap_matrix_dense(regions, abs_tol=pc.ABS_TOL):
    """n x n bool: is_fulldim(regions[i].intersect(regions[j])) for every pair (True on the diagonal).

    A pair of regions overlaps iff some pair of their member polytopes has a stacked H-representation with
    Chebyshev radius > abs_tol (Polytope.intersect reduces the stack, which keeps the set; Region.intersect
    keeps the pieces with chebR > abs_tol; ref :255-275, :815-830)."""
    n = len(regions)
    over = np.eye(n, dtype=bool)
    if n < 2:
        return over
    cells = _uniform_single_cells(regions) if solvers.default_solver == "hip" else None
    if cells is not None:
        from .batch import overlap_pairs
        return overlap_pairs(np.stack([c.A for c in cells]), np.stack([c.b for c in cells]), abs_tol=abs_tol).astype(bool)
    stacks, owner = [], []
    for i in range(n):
        li = regions[i].list_poly if isinstance(regions[i], pc.Region) else [regions[i]]
        for j in range(i):
            lj = regions[j].list_poly if isinstance(regions[j], pc.Region) else [regions[j]]
            for p in li:
                for q in lj:
                    if pc.is_fulldim(p) and pc.is_fulldim(q):
                        stacks.append(pc.Polytope(np.vstack([p.A, q.A]), np.hstack([p.b, q.b])))
                        owner.append((i, j))
    chunk = 65536
    for s0 in range(0, len(stacks), chunk):
        for (i, j), r in zip(owner[s0:s0 + chunk], pc._radii(stacks[s0:s0 + chunk])):
            if r > abs_tol:
                over[i, j] = over[j, i] = True
    return over


def are_disjoint(partition, check_all=False):
    """Return True if all Regions are (pairwise) disjoint, as Partition.are_disjoint does (ref :123-192):
    offending pairs are logged in the reference's order (i ascending, j < i ascending); without
    `check_all` only the first offending pair of the first offending region is reported."""
    regions = _regions_of(partition)
    over = overlap_matrix_dense(regions)
    ii, jj = np.nonzero(np.tril(over, -1))  # offending pairs, i ascending then j ascending
    seen = set()
    for i, j in zip(ii.tolist(), jj.tolist()):
        if not check_all and i in seen:
            continue  # the reference breaks out of the inner loop at the first offender of region i
        seen.add(i)
        logger.error("PPP is not a partition, regions: %d and: %d intersect each other.", i, j)
    return ii.size == 0


def compute_adj(partition, previous=None):
    """Adjacency matrix from scratch and its comparison with a previous one, as
    MetricPartition.compute_adj does (ref :244-306) -> (adj lil_matrix, ok)."""
    regions = _regions_of(partition)
    adj = sp.lil_matrix(adjacency_matrix_dense(regions).astype(float))
    ok = True
    if previous is not None:
        new, old = adj.toarray() != 0, sp.lil_matrix(previous).toarray() != 0
        for i, j in zip(*np.nonzero(new & ~old)):
            ok = False
            logger.error("PPP adjacency matrix is incomplete, missing: (%d, %d)", i, j)
        for i, j in zip(*np.nonzero(old & ~new)):
            ok = False
            logger.error("PPP adjacency matrix is incorrect, has 1 at: (%d, %d)", i, j)
        if not ok:
            logger.error("PPP had incorrect adjacency matrix.")
    return adj, ok


def find_adjacent_regions(partition):
    """Return region pairs that are spatially adjacent, as the reference does.

    @type partition: iterable container of L{Region} (anything with `.regions`, or a list)
    @rtype: scipy.sparse.lil_matrix (n x n, int8, ones on the diagonal)
    """
    regions = _regions_of(partition)
    return sp.lil_matrix(adjacency_matrix_dense(regions))
