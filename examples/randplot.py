#!/usr/bin/env python
"""BASELINE configs[0]: the plumbing case.  Random points in the unit square -> convex hull as an H-polytope ->
its vertices back -> minimal representation and Chebyshev ball, on the selected LP backend (no plotting).

    python examples/randplot.py [--points N] [--solver scipy|hip] [--seed S]

`--solver hip` needs an MI355X and the built libplp_hip.so; `scipy` runs anywhere.
"""
import argparse
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import polytope_amd.polytope as pc  # noqa: E402
from polytope_amd import solvers  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--points", type=int, default=10)
    ap.add_argument("--solver", default=None, choices=sorted(solvers.installed_solvers))
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args()
    if args.solver:
        solvers.default_solver = args.solver
    rng = np.random.default_rng(args.seed)
    cloud = rng.random((args.points, 2))
    hull = pc.qhull(cloud)
    if hull.A.size == 0:
        print(f"{args.points} points do not span the plane: empty hull")
        return
    corners = pc.extreme(hull)
    slim = pc.reduce(pc.Polytope(hull.A, hull.b))
    radius, centre = pc.cheby_ball(slim)
    print(f"backend          : {solvers.default_solver}")
    print(f"points / facets  : {args.points} / {hull.A.shape[0]} ({slim.A.shape[0]} after reduce)")
    print(f"hull vertices    : {len(corners)}")
    for v in corners[np.lexsort(corners.T[::-1])]:
        print(f"    ({v[0]:.6f}, {v[1]:.6f})")
    print(f"Chebyshev ball   : r = {radius:.6f} at ({centre[0]:.6f}, {centre[1]:.6f})")


if __name__ == "__main__":
    main()
