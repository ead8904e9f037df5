"""Adjacency of the regions of a partition -- mirror of the O(n^2) pair loop of the reference
(polytope/prop2partition.py:46-63, `find_adjacent_regions`), which calls `is_adjacent` on every
pair (i, j < i) and therefore issues one Chebyshev LP per pair of member polytopes
(polytope/polytope.py:1843-1866).  Here all pair LPs go to the device as ONE batch.

Only the adjacency computation is rebuilt; the `Partition` / `MetricPartition` containers of
the reference are plain Python bookkeeping around it.
"""
import numpy as np
import scipy.sparse as sp

from . import polytope as pc
from . import solvers


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
    if 2 * m > 64 or d > 8 or m < 1:
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


def find_adjacent_regions(partition):
    """Return region pairs that are spatially adjacent, as the reference does.

    @type partition: iterable container of L{Region} (anything with `.regions`, or a list)
    @rtype: scipy.sparse.lil_matrix (n x n, int8, ones on the diagonal)
    """
    regions = _regions_of(partition)
    return sp.lil_matrix(adjacency_matrix_dense(regions))
