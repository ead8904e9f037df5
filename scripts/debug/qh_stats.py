import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PLP_QH_STATS"] = "1"
import numpy as np
from scipy.spatial import ConvexHull
from polytope_amd import solvers
import polytope_amd.quickhull as Q
solvers.default_solver = "hip"
Q.quickhull(np.random.default_rng(0).standard_normal((1000, 3)))
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(200000, 4), (100000, 5), (20000, 6)]
for (N, d) in shapes:
    P = np.random.default_rng(N + d).standard_normal((N, d))
    for rep in range(2):
        np.random.seed(0)
        t = time.perf_counter(); A, b, V = Q.quickhull(P); tn = time.perf_counter() - t
    t = time.perf_counter(); ch = ConvexHull(P); ts = time.perf_counter() - t
    print("N=%d d=%d facets %d native %.4f s  scipy-qhull %.4f s" % (N, d, A.shape[0], tn, ts), flush=True)
