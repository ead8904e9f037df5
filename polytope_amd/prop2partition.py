"""Partitions of a polytopic set -- mirror of polytope/prop2partition.py of the reference: `find_adjacent_regions`
(:46-63), `Partition` (:68-228: container protocol, `is_partition`, `is_cover`, `are_disjoint`, `refines`,
`preserves`) and `MetricPartition` (:231-306: `compute_adj`).  Same names, constructor, attributes (`set`, and the
`regions` / `domain` / `adj` / `_elements` attributes the reference expects its subclasses to provide) and return values.

What differs is where the O(n^2) pair loops go.  The reference calls `is_adjacent` / `intersect` / `<=` on one
pair at a time, each a handful of LPs issued from Python (polytope/polytope.py:1843-1866, :815-830, :1032-1050).
Here every pair loop is ONE batch on the 'hip' backend:

    find_adjacent_regions, compute_adj   n(n-1)/2 stacked, abs_tol-inflated Chebyshev LPs formed on the device
    are_disjoint                         n(n-1)/2 stacked Chebyshev LPs (radius of the pairwise intersections)
    refines                              |self| x |other| stacked Chebyshev LPs screen the candidate supersets, then
                                         one region_diff search per candidate pair
    is_cover                             one region_diff batch of the domain against all member polytopes

Three places where the reference as written cannot run are given their evident meaning instead of its exception
(checked against the reference in tests/golden/make_golden.py, g16):
  * `is_cover` on a set that is not covered calls `logger.Error` (:110, an AttributeError): here the message is
    logged with `logger.error`, the warning is issued and False is returned;
  * `preserves` builds `set(other)` (:218) of Regions, which the reference's Region (it defines __eq__ without
    __hash__) does not allow: the Regions of this package hash by identity, so the difference is by identity;
  * `are_disjoint(fname=...)` saves figures through `Region.plot`; plotting is out of scope
    and the argument raises NotImplementedError.
"""
import logging
import warnings

import numpy as np
import scipy.sparse as sp

from . import polytope as pc
from . import solvers

logger = logging.getLogger(__name__)


def _regions_of(partition):
    return partition.regions if hasattr(partition, "regions") else list(partition)


def _members_of(regions):
    """Member polytopes of a list of Polytopes / Regions, flattened -> (members, first[n + 1]): region k owns
    members[first[k]:first[k + 1]] (none for an empty Region or an empty Polytope)."""
    members, first = [], [0]
    for r in regions:
        if isinstance(r, pc.Region):
            members += [p for p in r.list_poly if p.A.size]
        elif r.A.size:
            members.append(r)
        first.append(len(members))
    return members, np.asarray(first)


def _device_pairs_ok(members):
    """Can the pair kernels (csrc/plp_capi.hip: plp_adjacent_pairs / plp_overlap_pairs) take these polytopes?
    They stack two members per LP in registers: 2 * m_max <= 64 rows, one dimension d <= 16."""
    if solvers.default_solver != "hip":
        return False
    return _pair_shapes_ok(members)


def _pair_shapes_ok(members):
    if len(members) < 2:
        return False
    d = members[0].A.shape[1]
    return 1 <= d <= 16 and all(p.A.shape[1] == d and 1 <= p.A.shape[0] <= 32 for p in members)


def _members_key(regions):
    """identity of every element AND of every member polytope inside an element Region"""
    return tuple((id(r), tuple(id(p) for p in r.list_poly)) if isinstance(r, pc.Region) else id(r) for r in regions)


def _flat_members(regions, owner=None):
    """(members, first, pair kernels applicable) of `regions`; remembered on `owner` -- the object whose element list
    `regions` is -- keyed by the identity of EVERY element and of every member polytope inside an element Region, and the
    entry keeps the elements and members alive (an id cannot be handed out again while its object lives): an element
    replaced at the same length, or a member replaced inside an element (`reg.list_poly[k] = q`), gives another key.
    (Rows edited in place inside a member are caught one level down: polytope._table_of compares the packed rows' content.)"""
    if owner is not None and regions:
        key = _members_key(regions)
        hit = owner.__dict__.get("_p2p_flat")
        if hit is not None and hit[0] == key:
            return hit[1]
    members, first = _members_of(regions)
    out = (members, first, _pair_shapes_ok(members))
    if owner is not None and regions:
        owner.__dict__["_p2p_flat"] = (key, out, list(regions), list(members))
    return out


def _pair_list_device(regions, kind, abs_tol, diagonal=False, owner=None):
    """All member pairs of all regions as one batch of stacked Chebyshev LPs formed on the device, folded to the regions:
    (rows, cols) of the region pairs (i != j, both orders, sorted by row then column) for which SOME member pair is set
    (the `any` over member polytopes of is_adjacent, ref polytope.py:1843-1853, and of Region.intersect, :815-830);
    None when the pair kernels do not take the shapes (the callers then go pair by pair).  The members' rows are
    packed and uploaded once (polytope._table_of): a second call on the same regions -- the adjacency after the
    disjointness check, compute_adj against a previous matrix -- moves no input bytes; the n x n result stays on the
    device and only the indices of its nonzeros come back."""
    members, first, ok = _flat_members(regions, owner)
    if not ok or solvers.default_solver != "hip":
        return None
    from . import batch
    At, bt, mt = pc._table_of(members, owner=owner if len(members) == len(regions) else None).dev()
    fn = batch.adjacent_pairs if kind == "adjacent" else batch.overlap_pairs
    flat = len(members) == len(regions) and not np.any(np.diff(first) != 1)
    M = fn(At, bt, m=mt, abs_tol=abs_tol)
    if isinstance(M, np.ndarray):    # no torch in the process: the matrix came back through the host-pointer entry point
        def nonzero(M_):
            return np.argwhere(M_)
        if diagonal and flat:
            np.fill_diagonal(M, 1)
    else:
        torch = pc._torch_or_none()

        def nonzero(M_):
            return torch.nonzero(M_).cpu().numpy()
        if diagonal and flat:
            M.fill_diagonal_(1)
    if diagonal and flat:
        # (`diagonal`: the caller wants ones on the diagonal as well -- set on the device, so that the indices come back
        # sorted with them in place and no sort / unique runs on the host)
        nz = nonzero(M)
        return nz[:, 0], nz[:, 1], True
    nz = nonzero(M)     # member pairs, row-major order
    r, c = nz[:, 0], nz[:, 1]
    if not flat:
        owner = np.repeat(np.arange(len(regions)), np.diff(first))
        r, c = owner[r], owner[c]
        code = np.unique(r * len(regions) + c)
        r, c = code // len(regions), code % len(regions)
    off = r != c
    return (r[off], c[off], False) if diagonal else (r[off], c[off])


def _lil_from_pairs(n, r, c, dtype, diagonal=True):
    """lil_matrix with ones at (r, c) (sorted by row, then column) and on the diagonal (`diagonal`: not among the pairs
    yet), built from its row lists (the constructor from a dense array walks all n^2 entries)."""
    if diagonal:
        code = np.unique(np.concatenate([r * n + c, np.arange(n) * (n + 1)]))
        r, c = code // n, code % n
    L = sp.lil_matrix((n, n), dtype=dtype)
    ends = np.cumsum(np.bincount(r, minlength=n)).tolist()
    cols = c.tolist()
    one = np.dtype(dtype).type(1).item()
    rows, data = np.empty(n, dtype=object), np.empty(n, dtype=object)
    start = 0
    for k, e in enumerate(ends):
        rows[k] = cols[start:e]
        data[k] = [one] * (e - start)
        start = e
    L.rows, L.data = rows, data
    return L


def adjacency_matrix_dense(regions, abs_tol=pc.ABS_TOL):
    """n x n int8 adjacency (1 on the diagonal) -- is_adjacent(a, b, overlap=True) for every pair."""
    n = len(regions)
    adj = np.eye(n, dtype=np.int8)
    if n < 2:
        return adj
    got = _pair_list_device(regions, "adjacent", abs_tol)
    if got is not None:
        # the stacked, abs_tol-inflated pair LPs (polytope.py:1860-1866) are formed on the device
        adj[got[0], got[1]] = 1
        return adj
    ii, jj = np.tril_indices(n, -1)
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
    got = _pair_list_device(regions, "overlap", abs_tol)
    if got is not None:
        over[got[0], got[1]] = True
        return over
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


def touch_matrix(smalls, bigs, abs_tol=pc.ABS_TOL):
    """len(smalls) x len(bigs) bool: does some member of smalls[i] meet some member of bigs[j] in a set of Chebyshev
    radius > abs_tol -- the scan region_diff opens with (ref polytope.py:2148-2158), for every pair at once; None when
    the device pair kernels do not take the shapes."""
    if solvers.default_solver != "hip":
        return None
    m1, f1 = _members_of(smalls)
    m2, f2 = _members_of(bigs)
    if abs_tol != pc.ABS_TOL:
        return None
    got = pc._cross_touch(m1, m2)
    if got is None:
        return None
    # fold the member pairs to the elements (any member pair)
    n1, n2 = len(f1) - 1, len(f2) - 1
    out = np.zeros((n1, n2), dtype=bool)
    h1 = np.nonzero(f1[1:] > f1[:-1])[0]
    h2 = np.nonzero(f2[1:] > f2[:-1])[0]
    if h1.size and h2.size:
        out[np.ix_(h1, h2)] = np.maximum.reduceat(np.maximum.reduceat(got, f1[h1], axis=0), f2[h2], axis=1)
    return out


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
    adj = _adjacency_lil(regions, float)
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


def _adjacency_lil(regions, dtype, owner=None):
    n = len(regions)
    got = _pair_list_device(regions, "adjacent", pc.ABS_TOL, diagonal=True, owner=owner) if n >= 2 else None
    if got is not None:
        return _lil_from_pairs(n, got[0], got[1], dtype, diagonal=not got[2])
    return sp.lil_matrix(adjacency_matrix_dense(regions).astype(dtype))


def find_adjacent_regions(partition):
    """Return region pairs that are spatially adjacent, as the reference does.

    @type partition: iterable container of L{Region} (anything with `.regions`, or a list)
    @rtype: scipy.sparse.lil_matrix (n x n, int8, ones on the diagonal)
    """
    return _adjacency_lil(_regions_of(partition), np.int8, owner=partition if hasattr(partition, "__dict__") else None)


################################


def _all_polytopic(items):
    return all(isinstance(x, (pc.Polytope, pc.Region)) for x in items)


class Partition(pc._NoDeviceState):
    """Partition of a set (ref :68-228).

    An iterable container of sets over `Partition.set`; the members (`self.regions`) implement union / `__add__`,
    difference, intersection and `__le__`, so the builtin `set` can be used for discrete sets and Polytope / Region
    for polytopic ones.  As in the reference the constructor stores only `set`; `regions` (and `domain` for
    `is_cover`, `_elements` for `preserves`) are provided by the user or the subclass.
    """

    def __init__(self, domain=None):
        # `domain` rather than `set` to avoid shadowing the builtin (ref :85-91)
        self.set = domain

    def __len__(self):
        return len(self.regions)

    def __iter__(self):
        return iter(self.regions)

    def __getitem__(self, key):
        return self.regions[key]

    def is_partition(self):
        """Return True if Regions are pairwise disjoint and cover domain (ref :102-105)."""
        return self.is_cover() and self.are_disjoint()

    def is_cover(self):
        """Return True if Regions cover domain (ref :107-121): `domain <= union of the regions`.

        The reference accumulates the union with `+=`, i.e. union(check_convex=True), which merges members into convex
        pieces at O(n^2) envelope tests; that changes how the union is written, not the set.  For polytopic members the
        union is taken as the list of all member polytopes and the subset test (one region_diff search of each member
        of the domain against that list, then the volume of what is left: polytope.py:1032-1050) decides."""
        regions = list(self.regions)
        if _all_polytopic(regions) and isinstance(self.domain, (pc.Polytope, pc.Region)):
            members, _ = _members_of(regions)
            union = pc.Region(members)
        else:
            union = pc.Region()
            for region in regions:
                union += region
        if not self.domain <= union:
            msg = "partition does not cover domain."
            logger.error(msg)
            warnings.warn(msg)
            return False
        return True

    def are_disjoint(self, check_all=False, fname=None):
        """Return True if all Regions are disjoint (ref :123-192).

        For every offending pair the reference's report is logged: the two regions, the volume of their intersection
        and of their difference as a percentage of their mean volume.  Without `check_all` the scan of region i
        stops at its first offender (the reference's `break` leaves the inner loop only).

        @param check_all: report every offending pair
        @param fname: path prefix for the reference's debugging figures; plotting is out of scope here
        """
        logger.info("checking if PPP is a partition.")
        if fname:
            raise NotImplementedError("are_disjoint(fname=...): the figures need Region.plot, which is out of scope")
        regions = list(self.regions)
        over = overlap_matrix_dense(regions)
        ok = True
        for i, region in enumerate(regions):
            for j in np.nonzero(over[i, :i])[0].tolist():
                other = regions[j]
                isect = region.intersect(other)
                diff = region.diff(other)
                mean_volume = (region.volume + other.volume) / 2.0
                overlap = 100 * isect.volume / mean_volume
                non_overlap = 100 * diff.volume / mean_volume
                msg = "PPP is not a partition, regions: " + str(i) + " and: " + str(j) + " intersect each other.\n"
                msg += "Offending regions are:\n" + 10 * "-" + "\n"
                msg += str(region) + 10 * "-" + "\n" + str(other) + 10 * "-" + "\n"
                msg += "|cap| = " + str(overlap) + " %\n" + "|diff| = " + str(non_overlap) + "\n"
                logger.error(msg)
                ok = False
                if not check_all:
                    break
        return ok

    def refines(self, other):
        """Return True if each element is a subset of some element of `other` (ref :194-207).

        For polytopic elements on the 'hip' backend one batch of stacked Chebyshev LPs finds, for every element, the
        elements of `other` it meets at all; `small <= big` runs only for those.  An element that does not meet `big`
        is left whole by the difference (region_diff returns its minuend, polytope.py:2154-2158), so there
        `small <= big` is `small.volume < ABS_TOL`."""
        smalls, bigs = list(self), list(other)
        touch = None
        if smalls and bigs and _all_polytopic(smalls) and _all_polytopic(bigs):
            touch = touch_matrix(smalls, bigs)
            if solvers.default_solver == "hip":
                # the bounding boxes of all elements in ONE batch: `small <= big` reads them (polytope._inside_by_boxes)
                # before it forms a difference -- 1000 cells against two halves: 195 -> ms of array arithmetic
                members, _ = _members_of(smalls)
                todo = [p for p in members if p.bbox is None]
                if todo:
                    for p, box in zip(todo, pc._bbox_raw(todo)):
                        p.bbox = box
        for i, small in enumerate(smalls):
            found_superset = False
            for j, big in enumerate(bigs):
                if touch is not None and not touch[i, j]:
                    inside = bool(small.volume < pc.ABS_TOL)
                else:
                    inside = small <= big
                if inside:
                    found_superset = True
                    break
            if not found_superset:
                return False
        return True

    def preserves(self, other):
        """Does this partition respect `other` together with the complements of its elements (ref :209-228)?  Each element
        carries, as `.supersets`, the elements of `other` it is meant to lie inside; it must be a subset of every one of those
        and have an empty (zero-volume) intersection with every remaining element of `other`.
        Same calls in the same order as the reference makes them (`<=`, then `intersect` + truth value per remaining set,
        iterating a `set` difference), because both go through sampled volumes; only the bookkeeping is written differently."""
        every = set(other)
        for element in self._elements:
            inside = element.supersets
            if any(not (element <= big) for big in inside):
                return False
            outside = every.difference(inside)
            if any(bool(element.intersect(far)) for far in outside):
                return False
        return True


class MetricPartition(Partition):
    """Partition of a metric space, with the adjacency of its regions (ref :231-306).

    Two subsets are adjacent if the intersection of their closures is non-empty."""

    def compute_adj(self):
        """Update the adjacency matrix `self.adj` by checking all region pairs (polytope.is_adjacent on each, as one
        batch) and compare it with the previous one, if any.  -> True if the previous matrix was right (ref :244-306).
        """
        regions = list(self.regions)
        logger.info("computing adjacency from scratch...")
        adj = _adjacency_lil(regions, float)
        logger.info("...done computing adjacency.")
        ok = True
        if self.adj is not None:
            logger.info("checking previous adjacency...")
            new, old = adj.toarray(), sp.lil_matrix(self.adj).toarray()
            for i, j in zip(*np.nonzero(new)):
                if new[i, j] != old[i, j]:
                    ok = False
                    logger.error("PPP adjacency matrix is incomplete, missing: (" + str(i) + ", " + str(j) + ")")
            for i, j in zip(*np.nonzero(old)):
                if new[i, j] != old[i, j]:
                    ok = False
                    logger.error("PPP adjacency matrix is incorrect, has 1 at: (" + str(i) + ", " + str(j) + ")")
            if not ok:
                logger.error("PPP had incorrect adjacency matrix.")
            logger.info("done checking previous adjacency.")
        else:
            logger.info("no previous adjacency found: skip verification.")
        self.adj = adj
        return ok
