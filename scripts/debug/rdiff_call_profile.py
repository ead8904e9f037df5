"""region_diff at config 4 (fixture g12), warm: cProfile of the object-level call (what surrounds the library search)."""
import cProfile, itertools, os, pstats, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import polytope_amd as pc
from conftest import load_golden
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
g = load_golden("g12_config4.npz")
P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
f = lambda: pc.polytope.region_diff(P.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
for _ in range(3):
    t0 = time.perf_counter(); D = f(); print("%.2f ms, %d pieces" % ((time.perf_counter() - t0) * 1e3, len(D)))
pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(24)
