import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from polytope_amd import solvers
import polytope_amd.quickhull as Q
solvers.default_solver = "hip"
Q.quickhull(np.random.default_rng(0).standard_normal((1000, 3)))
N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 5)
P = np.random.default_rng(N + d).standard_normal((N, d))
for rep in range(3):
    np.random.seed(0)
    t = time.perf_counter(); Q.quickhull(P); print("rep", rep, "%.4f s" % (time.perf_counter() - t))
np.random.seed(0)
pr = cProfile.Profile(); pr.enable(); Q.quickhull(P); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
