"""GPU: does the facet graph on the device (PLP_QH_DEVICE_TAIL=1, csrc/plp_quickhull_dev.hip) win anywhere?  Many-facet
hulls (points on / near a sphere, d = 3..7) against the default (host facet graph + device kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from polytope_amd import solvers
import polytope_amd.quickhull as Q
solvers.default_solver = "hip"
rng = np.random.default_rng(0)
for (N, d, kind) in [(200000, 3, "sphere"), (50000, 4, "sphere"), (20000, 5, "sphere"), (6000, 6, "sphere"), (2000, 7, "sphere"),
                     (1000000, 3, "gauss"), (300000, 5, "gauss"), (100000, 6, "gauss"), (30000, 7, "gauss")]:
    P = rng.standard_normal((N, d))
    if kind == "sphere":
        P /= np.linalg.norm(P, axis=1)[:, None]
    res = {}
    for mode in ("host", "device", "host", "device"):
        os.environ.pop("PLP_QH_DEVICE_TAIL", None)
        if mode == "device":
            os.environ["PLP_QH_DEVICE_TAIL"] = "1"
        np.random.seed(0)
        t = time.perf_counter()
        try:
            A, b, V = Q.quickhull(P)
            res[mode] = (time.perf_counter() - t, A.shape[0])
        except Exception as e:
            res[mode] = (float("nan"), repr(e)[:60])
    print("N=%7d d=%d %-6s facets %s  host graph %.4f s  device graph %.4f s  ratio %.2f" % (
        N, d, kind, res["host"][1], res["host"][0], res["device"][0], res["device"][0] / res["host"][0]), flush=True)
