import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.spatial import ConvexHull
from polytope_amd import solvers
import polytope_amd.quickhull as Q
solvers.default_solver = "hip"
for (N, d) in [(1000, 3), (200000, 4), (1000000, 3), (1000000, 2), (100000, 5), (20000, 6)]:
    P = np.random.default_rng(N + d).standard_normal((N, d))
    out = {}
    for native in (True, False, True):
        if not native and (d >= 5): continue
        Q._NATIVE_LOOP = native
        np.random.seed(0)
        t = time.perf_counter()
        A, b, V = Q.quickhull(P)
        out[native] = (A, b, V, time.perf_counter() - t)
    t = time.perf_counter(); ch = ConvexHull(P); ts = time.perf_counter() - t
    same = np.array_equal(np.sort(out[True][2], axis=0), np.sort(P[np.unique(ch.vertices)], axis=0))
    msg = "N=%d d=%d facets %d native %.4f s" % (N, d, out[True][0].shape[0], out[True][3])
    if False in out:
        msg += "  python-graph %.4f s  identical rows: %s" % (out[False][3], np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][1], out[False][1]))
    print(msg, " scipy-qhull %.4f s  same vertex set: %s" % (ts, same), flush=True)
