"""is_subset(200 cells, 1000 cells) and Partition.refines, warm: cProfile."""
import cProfile, itertools, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pc
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
f = lambda: pc.is_subset(pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells]))
for _ in range(3):
    t0 = time.perf_counter(); r = f(); print("%.2f ms" % ((time.perf_counter() - t0) * 1e3), r)
pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
coarse = pc.Partition(); coarse.regions = [pc.Region([pc.box2poly([[i / 2, (i + 1) / 2]] + [[0, 1]] * 3)]) for i in range(2)]
fine = pc.Partition(); fine.regions = [pc.Region([c]) for c in cells]
g = lambda: fine.refines(coarse)
for _ in range(3):
    t0 = time.perf_counter(); r = g(); print("refines %.2f ms" % ((time.perf_counter() - t0) * 1e3), r)
pr = cProfile.Profile(); pr.enable(); g(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(26)
