import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polytope_amd.polytope as pc
from polytope_amd import solvers, synth
solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))
import cProfile, pstats
for rep in range(2):
    R = pc.Region([c.copy() for c in cells])
    t = time.perf_counter(); I = R.intersect(P.copy()); t1 = time.perf_counter() - t
    print("Region(1000).intersect(P): %.3f s, pieces %d" % (t1, len(I) if isinstance(I, pc.Region) else 1), flush=True)
R = pc.Region([c.copy() for c in cells])
pr = cProfile.Profile(); pr.enable(); I = R.intersect(P.copy()); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
t = time.perf_counter(); s = pc.is_subset(pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells])); print("is_subset(200 cells, 1000 cells): %.3f s -> %s" % (time.perf_counter() - t, s))
