import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd import synth
from conftest import load_golden
dev = torch.device("cuda:0")
def run(A, b, X, tol, m=None, region=True):
    out = {}
    for v in ("0", "1"):
        os.environ["PLP_CONTAINS_MFMA"] = v
        out[v] = pa.contains_batch(A, b, X, tol, m=m, region=region)
    return out
# small shapes, per-polytope mode, ragged m, several d
rng = np.random.default_rng(1)
for (P, m, d, N) in [(8, 16, 6, 4096), (5, 16, 3, 1000), (7, 40, 7, 3000), (3, 10, 2, 257), (4, 64, 16, 2000), (6, 20, 11, 1500)]:
    A, b, X = synth.containment_workload(P, N, d=d, m=max(m, 2 * d), seed=3)
    A, b = A[:, :m], b[:, :m]
    ms = rng.integers(max(1, m - 5), m + 1, P).astype(np.int32)
    for region in (True, False):
        o = run(A, b, X, 1e-7, m=ms, region=region)
        print((P, m, d, N), "region" if region else "per-polytope", "equal:", np.array_equal(o["0"], o["1"]), "inside", int(np.asarray(o["1"]).sum()))
# boundary points: g4
g = load_golden("g4_contains.npz")
print({k: v.shape for k, v in g.items() if hasattr(v, "shape")})
# points exactly on facets: box corners / faces
P, d = 16, 3
A = np.tile(np.vstack([np.eye(d), -np.eye(d)]), (P, 1, 1)); b = np.ones((P, 2 * d))
X = rng.integers(-2, 3, (d, 5000)).astype(float) * 0.5
for tol in (0.0, 1e-7, 0.01):
    o = run(A, b, X, tol, region=False)
    print("lattice points on faces tol", tol, "equal:", np.array_equal(o["0"], o["1"]), int(np.asarray(o["1"]).sum()))
# C3 timing
P, N, d, m = 10000, 1000000, 6, 16
A, b, X = synth.containment_workload(P, N, d=d, m=m, seed=0)
At, bt, Xt = (torch.as_tensor(v).to(dev) for v in (A, b, X))
res = {}
for v in ("0", "1"):
    os.environ["PLP_CONTAINS_MFMA"] = v
    r = pa.contains_batch(At, bt, Xt, 1e-7); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): r = pa.contains_batch(At, bt, Xt, 1e-7)
    torch.cuda.synchronize()
    res[v] = (r.cpu().numpy(), (time.perf_counter() - t) / 3)
print("C3: valu %.2f ms, mfma %.2f ms, equal: %s, inside %d" % (res["0"][1] * 1e3, res["1"][1] * 1e3, np.array_equal(res["0"][0], res["1"][0]), int(res["1"][0].sum())))
