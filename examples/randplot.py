#!/usr/bin/env python
"""Sample N points in the unit square, compute their hull and its extreme points -- the reference's
examples/randplot.py without the plot (BASELINE configs[0]), on whichever backend is selected.

  Usage: randplot.py [N] [solver]      solver: hip (default, needs an MI355X) or scipy
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polytope_amd.polytope as polytope  # noqa: E402
from polytope_amd import solvers  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    if len(sys.argv) > 2:
        solvers.default_solver = sys.argv[2]
    V = np.random.rand(N, 2)
    print("Sampled " + str(N) + " points:")
    print(V)
    P = polytope.qhull(V)
    print("Computed the convex hull:")
    print(P)
    V_min = polytope.extreme(P)
    print("which has extreme points:")
    print(V_min)
    if P.A.size:
        P = polytope.reduce(polytope.Polytope(P.A, P.b))
        r, xc = polytope.cheby_ball(P)
        print("Chebyshev ball: r = %.6f at %s (solver: %s)" % (r, np.asarray(xc).ravel(), solvers.default_solver))
