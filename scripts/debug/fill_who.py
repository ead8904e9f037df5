"""Which polytopes reach union(check_convex) without a cached Chebyshev ball (STRICT_REFERENCE_QUIRKS: one small LP batch per step)?"""
import itertools, os, sys, traceback
from collections import Counter
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pc
import polytope_amd.polytope as pp
from polytope_amd import synth
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))
f = lambda: pc.Region([c.copy() for c in cells]).intersect(P.copy())
f(); f()
orig = pp._cheby_fill
stats = []
def spy(polys):
    todo = [p for p in polys if not (p._chebXc is not None and p._chebR is not None) and not pp.is_empty(p)]
    if todo:
        st = traceback.extract_stack()
        stats.append((len(todo), tuple((q.A.shape, bool(q.minrep), q.fulldim) for q in todo[:2]), st[-2].name + "<" + st[-3].name))
    return orig(polys)
pp._cheby_fill = spy
f()
print(Counter((s[0], s[2]) for s in stats).most_common(8))
print(stats[:4])
