"""cProfile of region_diff at config 4 (library search) -- where the Python around the search goes."""
import sys, os, itertools, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polytope_amd.polytope as pc
from polytope_amd import solvers
from conftest import load_golden
solvers.default_solver = "hip"
g = load_golden("g12_config4.npz")
shape = tuple(int(v) for v in g["c4_shape"])
cells = [pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]) for idx in itertools.product(*[range(n) for n in shape])]
P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
def run():
    for c in cells: c._chebR = c._chebXc = None; c.fulldim = None
    return pc.region_diff(P.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
os.environ["PLP_RDIFF_STATS"] = "1"
for _ in range(3):
    t = time.perf_counter(); D = run(); print("pieces", len(D), "%.4f s" % (time.perf_counter() - t))
os.environ.pop("PLP_RDIFF_STATS")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
