"""cProfile of Region(1000).intersect(P) and is_subset(Region(200), Region(1000)) at d = 4, warm, STRICT_REFERENCE_QUIRKS on / off."""
import cProfile, itertools, os, pstats, sys
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
which = sys.argv[1] if len(sys.argv) > 1 else "subset"
g = (lambda: pc.is_subset(pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells]))) if which == "subset" else \
    (lambda: pc.Region([c.copy() for c in cells]).intersect(P.copy()))
for _ in range(3):
    g()
for quirks in (True, False):
    pp.STRICT_REFERENCE_QUIRKS = quirks
    g()
    pr = cProfile.Profile(); pr.enable(); g(); pr.disable()
    print("==== quirks", quirks)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(26)
