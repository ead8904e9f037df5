"""Wall time of single Python-level calls on the 'hip' backend (numpy in, numpy out): what a loop over polytopes pays."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import polytope_amd as pa
import polytope_amd.polytope as pc
from polytope_amd import solvers
from polytope_amd.synth import random_hpolytopes
solvers.default_solver = "hip"
def t(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
for (m, d) in [(16, 3), (12, 4), (32, 6)]:
    A, b = random_hpolytopes(64, m, d, seed=5, stream=0, bounded=True)
    print("(%d,%d): reduce_batch B=1 %.1f us, B=64 %.1f us | cheby_ball_batch B=1 %.1f us | pc.reduce(Polytope) %.1f us | Polytope.intersect %.1f us"
          % (m, d, t(lambda: pa.reduce_batch(A[:1], b[:1])), t(lambda: pa.reduce_batch(A, b)), t(lambda: pa.cheby_ball_batch(A[:1], b[:1])),
             t(lambda: pc.reduce(pc.Polytope(A[0], b[0]))), t(lambda: pc.Polytope(A[0], b[0]).intersect(pc.Polytope(A[1], b[1])))))
# one solvers.lpsolve call (the plug-in boundary): origin feasible / origin infeasible (phase 1)
rng = np.random.default_rng(1)
G = rng.standard_normal((16, 3)); G /= np.linalg.norm(G, axis=1)[:, None]
c = rng.standard_normal(3)
h1 = 1.0 + rng.random(16)
h2 = h1 + G @ np.array([3.0, -2.0, 1.0])          # same polytope moved away from the origin
print("solvers.lpsolve: origin feasible %.1f us, phase 1 needed %.1f us" % (t(lambda: solvers.lpsolve(c, G, h1)), t(lambda: solvers.lpsolve(c, G, h2))))
