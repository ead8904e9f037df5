"""Round-4 timings of the object-level calls at C4 size: Region(1000).intersect(P), is_subset(200 cells,
1000 cells), find_adjacent_regions first / second call, with the H2D bytes each call moved (batch.h2d_bytes)."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polytope_amd as pc
from polytope_amd import synth, batch
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))

def timed(label, fn, reps=3):
    for r in range(reps):
        h0 = batch.h2d_bytes
        t = time.perf_counter(); out = fn(); dt = time.perf_counter() - t
        print("%-44s %8.2f ms   h2d %9d B" % (label + " #%d" % r, dt * 1e3, batch.h2d_bytes - h0), flush=True)
    return out

timed("Region(1000).intersect(P)", lambda: pc.Region([c.copy() for c in cells]).intersect(P.copy()))
pc.polytope._hull_memo.clear(); pc.polytope._convex_memo.clear()
timed("  (memos cleared before #0)", lambda: pc.Region([c.copy() for c in cells]).intersect(P.copy()), reps=2)
timed("is_subset(200 cells, 1000 cells)", lambda: pc.is_subset(pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells])))
part = pc.MetricPartition(pc.box2poly([[0, 1]] * 4)); part.regions = [pc.Region([c]) for c in cells]; part.adj = None
timed("find_adjacent_regions(1000 regions)", lambda: pc.find_adjacent_regions(part), reps=4)
timed("are_disjoint(1000 regions)", lambda: part.are_disjoint(), reps=2)
coarse = pc.Partition(); coarse.regions = [pc.Region([pc.box2poly([[i / 2, (i + 1) / 2]] + [[0, 1]] * 3)]) for i in range(2)]
fine = pc.Partition(); fine.regions = part.regions
timed("Partition(1000).refines(Partition(2))", lambda: fine.refines(coarse), reps=2)
