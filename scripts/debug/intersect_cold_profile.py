"""cProfile of Region(1000 cells).intersect(P) with the union memos emptied (the first call of a process)."""
import os, sys, time, itertools, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polytope_amd as pc
from polytope_amd import synth
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))
pc.Region([c.copy() for c in cells[:50]]).intersect(P.copy())     # load the kernels
pc.polytope._hull_memo.clear(); pc.polytope._convex_memo.clear()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
I = pc.Region([c.copy() for c in cells]).intersect(P.copy())
pr.disable(); print("cold: %.1f ms, pieces %d" % ((time.perf_counter() - t) * 1e3, len(I)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
