"""union(P, Q, check_convex=True) of two adjacent boxes / two random polytopes, warm: which device calls it makes."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pc
pc.solvers.default_solver = "hip"
P0 = pc.box2poly([[0, 1], [0, 1], [0, 1]]); Q0 = pc.box2poly([[1, 2], [0, 1], [0, 1]]); R0 = pc.box2poly([[1, 2], [1, 2], [0, 1]])
def f():
    pc.polytope._hull_memo.clear(); pc.polytope._convex_memo.clear()
    return pc.union(P0.copy(), Q0.copy(), check_convex=True), pc.union(P0.copy(), R0.copy(), check_convex=True)
for _ in range(3):
    t0 = time.perf_counter(); f(); print("%.3f ms for a convex and a non-convex pair" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): f()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(26)
