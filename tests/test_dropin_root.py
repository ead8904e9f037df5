"""`import polytope_amd as polytope` as the whole switch: the reference's root-level names
(polytope/__init__.py:35-44) and its Partition / MetricPartition classes (polytope/prop2partition.py:68-306), replayed
against the vectors generated from the reference -- g7 (its own tests' known answers), g9 (pair matrices) and g16
(class-level outputs: is_cover / are_disjoint / is_partition / refines / preserves / compute_adj) -- once per backend:
'scipy' on the CPU, 'hip' (marked gpu) with every LP on the HIP kernels.
"""
import logging
import warnings

import numpy as np
import pytest

from conftest import load_golden

import test_python_api as api

ROOT_NAMES = ("Polytope Region is_empty is_fulldim is_convex is_adjacent is_subset reduce separate box2poly "
              "cheby_ball bounding_box envelope extreme qhull is_inside union mldivide intersect volume "
              "Partition MetricPartition find_adjacent_regions").split()


@pytest.fixture(params=["scipy", pytest.param("hip", marks=pytest.mark.gpu)])
def pc(request):
    import polytope_amd as pc   # the package root, not the submodule
    old = pc.solvers.default_solver
    if request.param == "hip":
        assert "hip" in pc.solvers.installed_solvers, "HIP backend not installed on a GPU box"
    pc.solvers.default_solver = request.param
    yield pc
    pc.solvers.default_solver = old


def test_root_exports_the_reference_names():
    """Every name of polytope/__init__.py:35-44 on the hot path; grid_region / projection are out of scope and absent
    (not stubbed); the submodules are reachable as in the reference (polytope.polytope, polytope.solvers, ...)."""
    import polytope_amd as pc
    assert [n for n in ROOT_NAMES if not hasattr(pc, n)] == []
    assert not hasattr(pc, "grid_region") and not hasattr(pc, "projection")
    assert pc.polytope.Polytope is pc.Polytope and pc.prop2partition.Partition is pc.Partition
    assert pc.solvers.lpsolve and pc.quickhull.quickhull and isinstance(pc.__version__, str)
    assert issubclass(pc.MetricPartition, pc.Partition)


def test_known_answers_through_root_names(pc):
    """g7: the numbers the reference's own tests pin (polytope_test.py), through the package root."""
    api.test_known_answers(pc)
    api.test_contains_semantics(pc)


def _cells(pc, g, name):
    d = int(g[name + "_d"])

    def polys(Ab, ms):
        out = []
        for row, m in zip(Ab, ms):
            M = row[:int(m) * (d + 1)].reshape(int(m), d + 1)
            out.append(pc.Polytope(M[:, :d], M[:, d], normalize=False))
        return out
    members = polys(g[name + "_Ab"], g[name + "_m"])
    owner = g[name + "_owner"]
    regions = [pc.Region([p for p, o in zip(members, owner) if o == k]) for k in range(int(owner.max()) + 1)]
    dom = polys(g[name + "_dom_Ab"], g[name + "_dom_m"])
    return (dom[0] if len(dom) == 1 else pc.Region(dom)), regions


def _partition(pc, g, name, cls="MetricPartition"):
    domain, regions = _cells(pc, g, name)
    part = getattr(pc, cls)(domain)
    assert part.set is domain and not hasattr(part, "regions")   # the constructor stores `set` only (ref :85-91)
    part.domain, part.regions, part.adj = domain, regions, None
    return part


def test_partition_container_protocol(pc):
    g = load_golden("g16_partition.npz")
    part = _partition(pc, g, "grid2", "Partition")
    assert len(part) == 12 and list(part) == part.regions and part[3] is part.regions[3]
    bare = pc.MetricPartition()
    assert bare.set is None
    with pytest.raises(AttributeError):
        bare.compute_adj()        # `regions` / `adj` come from the subclass, as in the reference
    with pytest.raises(AttributeError):
        pc.Partition(pc.box2poly([[0, 1]])).is_cover()


@pytest.mark.parametrize("name", ["grid2", "grid2_hole", "grid2_coarse", "grid2_cols", "grid3", "grid3_slabs", "rand2",
                                  "tri2", "tri2_squares", "overlap_cover", "slab4"])
def test_partition_classes_against_the_reference(pc, name, caplog):
    g = load_golden("g16_partition.npz")
    part = _partition(pc, g, name)
    n = len(part)
    # is_cover (:107-121).  Where the set is not covered the reference dies on `logger.Error`; the build logs, warns
    # and returns False.
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cover = part.is_cover()
    assert cover == bool(g[name + "_cover"])
    assert bool(g[name + "_cover_raised"]) == (not cover)
    assert any("does not cover" in str(x.message) for x in w) == (not cover)
    # are_disjoint (:123-192), with the report of the offending pairs
    with caplog.at_level(logging.ERROR, logger="polytope_amd.prop2partition"):
        caplog.clear()
        disjoint = part.are_disjoint()
        first = [r for r in caplog.records if "intersect each other" in r.getMessage()]
        caplog.clear()
        assert part.are_disjoint(check_all=True) == disjoint
        every = [r for r in caplog.records if "intersect each other" in r.getMessage()]
    assert disjoint == bool(g[name + "_disjoint"])
    assert (len(first) == 0) == disjoint and len(every) >= len(first)
    if not disjoint:
        assert "|cap| = " in every[0].getMessage() and "|diff| = " in every[0].getMessage()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert part.is_partition() == (cover and disjoint)
    with pytest.raises(NotImplementedError):
        part.are_disjoint(fname="/tmp/figure_")
    # compute_adj (:244-306): the matrix, the verdict on the previous one, self.adj updated
    assert part.compute_adj() == bool(g[name + "_adj_ok"][0])
    assert np.array_equal(part.adj.toarray(), g[name + "_adj"])
    assert part.compute_adj()
    wrong = part.adj.copy()
    wrong[0, n - 1] = 0.0 if wrong[0, n - 1] else 1.0
    part.adj = wrong
    assert part.compute_adj() == bool(g[name + "_adj_ok"][1])
    assert np.array_equal(part.adj.toarray(), g[name + "_adj"])
    # find_adjacent_regions (:46-63)
    far = pc.find_adjacent_regions(part)
    assert far.dtype == np.int8 and np.array_equal(far.toarray(), g[name + "_far"])


def test_partition_refines_and_preserves(pc):
    g = load_golden("g16_partition.npz")
    for pair, want in zip(g["refines_pairs"], g["refines"]):
        a, b = str(pair).split(">")
        assert _partition(pc, g, a, "Partition").refines(_partition(pc, g, b, "Partition")) == bool(want), pair
    # builtin sets, as the class docstring allows (ref :71-79)
    fine, coarse = pc.Partition(), pc.Partition()
    fine.regions, coarse.regions = [{1}, {2}, {3, 4}], [{1, 2}, {3, 4, 5}]
    assert fine.refines(coarse) and not coarse.refines(fine)
    assert not hasattr(g, "preserves_ref_hashable") and not bool(g["preserves_ref_hashable"])
    for case, want in zip(g["preserves_cases"], g["preserves"]):
        names, shift = str(case).split("+")
        a, b = names.split(">")
        pf, pb = _partition(pc, g, a, "Partition"), _partition(pc, g, b, "Partition")
        nb = len(pb)
        for item in pf.regions:
            xc = item.list_poly[0].chebXc
            k = [t for t, big in enumerate(pb.regions) if xc in big][0]
            item.supersets = [pb.regions[(k + int(shift)) % nb]]
        pf._elements = pf.regions
        assert pf.preserves(pb) == bool(want), case


@pytest.mark.parametrize("name", ["grid2", "grid3", "rand2", "rand3"])
def test_g9_pair_matrices_through_the_classes(pc, name):
    """g9: the reference's own pair loops (is_fulldim(region.intersect(other)), is_adjacent) on single-cell regions,
    now through MetricPartition / find_adjacent_regions of the package root."""
    g = load_golden("g9_overlap.npz")
    cells = [pc.Region([pc.Polytope(A, b)]) for A, b in zip(g[name + "_A"], g[name + "_b"])]
    part = pc.MetricPartition(pc.Region([c.list_poly[0] for c in cells]))
    part.regions, part.adj = cells, None
    off = g[name + "_over"] & ~np.eye(len(cells), dtype=bool)
    assert part.are_disjoint() == (not off.any())
    assert part.compute_adj() and np.array_equal(part.adj.toarray() != 0, g[name + "_adj"] != 0)
    assert np.array_equal(pc.find_adjacent_regions(part).toarray(), g[name + "_adj"])
    assert pc.is_adjacent(cells[0], cells[1]) == bool(g[name + "_adj"][0, 1])


def test_str_of_polytope_and_region(pc):
    """`print(p)`: the reference's rendering (polytope.py:150-176, :711-721), which its own test_polytope_str pins."""
    g = load_golden("g16_partition.npz")
    got = [str(pc.Polytope(np.array([[1]]), np.array([1])))]
    for box in ([[0, 1]], [[0, 1], [0, 2]], [[0, 1], [0, 2], [0, 3]]):
        got.append(str(pc.box2poly(box)))
    got.append(str(pc.Region([pc.box2poly([[0, 1], [0, 2]]), pc.box2poly([[1, 2], [0, 2]])])))
    got.append(str(pc.Polytope(np.array([[0.6, -0.8], [-1.0, 0.0], [0.0, 1.0]]), np.array([1.25, 0.5, 3.0]))))
    assert got == [str(v) for v in g["str_cases"]]
    assert got[0] == "Single polytope \n  [[1.]] x <= [[1.]]\n"
