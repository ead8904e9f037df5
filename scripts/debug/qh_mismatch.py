import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.spatial import ConvexHull
import polytope_amd.quickhull as Q
from polytope_amd import solvers
solvers.default_solver = "hip"
targets = {724, 1074, 1672, 2368}
rng = np.random.default_rng(77)
for trial in range(2400):
    d = int(rng.integers(2, 6))
    N = int(rng.integers(d + 2, 60000 if d < 4 else (6000 if d == 4 else 600)))
    P = rng.standard_normal((N, d)) if trial % 2 else rng.random((N, d))
    if trial % 5 == 0: P[N // 2:] = P[:N - N // 2]
    if trial % 7 == 0: P = np.round(P * 4) / 4
    if trial not in targets: continue
    res = {}
    for key, native, tail in (("native", True, None), ("native_notail", True, "0"), ("python", False, None)):
        Q._NATIVE_LOOP = native
        if tail is not None: os.environ["PLP_QH_HOST_TAIL"] = tail
        else: os.environ.pop("PLP_QH_HOST_TAIL", None)
        np.random.seed(trial)
        res[key] = Q.quickhull(P)
    ref = np.unique(P[np.unique(ConvexHull(P).vertices)], axis=0)
    for key, (A, b, V) in res.items():
        Vu = np.unique(V, axis=0)
        print(trial, d, N, key, "facets", A.shape[0], "verts", V.shape[0], "scipy verts", ref.shape[0],
              "vertex set == scipy:", Vu.shape == ref.shape and np.array_equal(Vu[np.lexsort(Vu.T[::-1])], ref[np.lexsort(ref.T[::-1])]),
              "max violation", float(np.max(A @ P.T - b[:, None])))
    print("  native == python rows:", res["native"][0].shape == res["python"][0].shape and np.array_equal(res["native"][0], res["python"][0]),
          " notail == python:", res["native_notail"][0].shape == res["python"][0].shape and np.array_equal(res["native_notail"][0], res["python"][0]))
