import os, sys, time, itertools, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polytope_amd.polytope as pc
from polytope_amd import solvers
solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
pc.is_subset(pc.Region([c.copy() for c in cells[:50]]), pc.Region([c.copy() for c in cells]))
a, b = pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells])
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter(); s = pc.is_subset(a, b); dt = time.perf_counter() - t; pr.disable()
print("is_subset(200, 1000): %.3f s -> %s" % (dt, s))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
