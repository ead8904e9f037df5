"""Region(1000 cells).intersect(P) and is_subset(Region(200), Region(1000)) at d = 4, warm, with STRICT_REFERENCE_QUIRKS on / off."""
import itertools, os, sys, time
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
g = lambda: pc.is_subset(pc.Region([c.copy() for c in cells[:200]]), pc.Region([c.copy() for c in cells]))
for quirks in (True, False, True):
    pp.STRICT_REFERENCE_QUIRKS = quirks
    for name, fn in (("Region(1000).intersect(P)", f), ("is_subset(200, 1000)", g)):
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        print("quirks %-5s %-28s first %.1f ms, then min %.1f / median %.1f ms" % (quirks, name, ts[0], min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]))
