"""Timing of region_diff at C4 (library search vs host loop) + equality of the two."""
import sys, os, itertools, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polytope_amd.polytope as pc
from polytope_amd import solvers, batch
from conftest import load_golden
solvers.default_solver = "hip"
g = load_golden("g12_config4.npz")
shape = tuple(int(v) for v in g["c4_shape"])
cells = [pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]) for idx in itertools.product(*[range(n) for n in shape])]
P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
orig = batch.region_diff_search
stats = {}
def spy(*a, **k):
    t = time.perf_counter()
    out = orig(*a, **k)
    stats["search_s"] = time.perf_counter() - t
    stats.update(out[1])
    return out
batch.region_diff_search = spy
res = {}
for native in (True, False, True):
    pc._RDIFF_NATIVE = native
    for c in cells: c._chebR = c._chebXc = None; c.fulldim = None
    t = time.perf_counter()
    D = pc.region_diff(P.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
    dt = time.perf_counter() - t
    res[native] = D
    print("native" if native else "host loop", "pieces", len(D), "time %.4f s" % dt, stats if native else "")
a, b = res[True].list_poly, res[False].list_poly
print("same pieces:", len(a) == len(b) and all(x.A.shape == y.A.shape and np.array_equal(x.A, y.A) and np.array_equal(x.b, y.b) for x, y in zip(a, b)))
# the small config of round 1 (P r~0.1, 500 cells)
from polytope_amd import synth
A, bb = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P2 = pc.Polytope(A[0], 0.1 * bb[0] + A[0] @ (0.5 * np.ones(4)))
index = np.array(list(itertools.product(*[range(n) for n in shape])))
half = [c for c, idx in zip(cells, index) if idx[0] < 5]
for native in (True, False, True):
    pc._RDIFF_NATIVE = native
    t = time.perf_counter()
    D = pc.region_diff(P2.copy(), pc.Region(half))
    print("small config:", "native" if native else "host loop", "pieces", len(D), "time %.4f s" % (time.perf_counter() - t), stats if native else "")
