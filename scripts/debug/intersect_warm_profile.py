"""Region(1000).intersect(P), warm: cProfile of one call (where the ~70 ms go)."""
import cProfile, itertools, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pc
from polytope_amd import synth
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))
f = lambda: pc.Region([c.copy() for c in cells]).intersect(P.copy())
for _ in range(3):
    t0 = time.perf_counter(); f(); print("%.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
