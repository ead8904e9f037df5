"""region_diff at d >= 5: a polytope minus random overlapping boxes; library search with the one-LP-per-wavefront gather
kernel against PLP_RDIFF_WIDE=0 (lane groups for d <= 8, LDS engine beyond)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import polytope_amd.polytope as pc
from polytope_amd import solvers
solvers.default_solver = "hip"
for d, n in [(6, 12), (8, 10), (10, 8), (5, 30)]:
    rng = np.random.default_rng(3)
    cen = rng.random((n, d)); hw = rng.uniform(0.15, 0.4, (n, d))
    cells = [pc.box2poly(np.c_[c - w, c + w].tolist()) for c, w in zip(cen, hw)]
    A = rng.standard_normal((3 * d, d)); A /= np.linalg.norm(A, axis=1)[:, None]
    P = pc.Polytope(A, 0.4 * (1 + rng.random(3 * d)) + A @ (0.5 * np.ones(d)))
    for env in ("0", "1"):
        os.environ["PLP_RDIFF_WIDE"] = env
        try:
            D = pc.region_diff(P.copy(), pc.Region([c.copy() for c in cells]))
            os.environ["PLP_RDIFF_STATS"] = "1"
            t = time.perf_counter(); D = pc.region_diff(P.copy(), pc.Region([c.copy() for c in cells])); dt = time.perf_counter() - t
            os.environ.pop("PLP_RDIFF_STATS")
            print("d=%d n=%d PLP_RDIFF_WIDE=%s: %.4f s, %d pieces" % (d, n, env, dt, len(D) if isinstance(D, pc.Region) else 1), flush=True)
        except IndexError:
            print("d=%d n=%d: IndexError (as the reference)" % (d, n))
