#!/usr/bin/env python3
"""Soak run on the GPU box (not part of the test suite): long fuzz of the LP kernels against the oracle with
several seeds, with and without the forced hand-over paths, and quickhull against scipy.spatial.ConvexHull on
many random inputs.  Usage: gpurun -- 'python scripts/soak.py [trials]'"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
trials = sys.argv[1] if len(sys.argv) > 1 else "1500"
bad = 0
for seed, extra in [("11", {}), ("12", {}), ("13", {"PLP_REDUCE_RETRY_ALL": "1", "PLP_CHEBY_RETRY_ALL": "1"}),
                    ("14", {"PLP_REDUCE_1ROW": "1", "PLP_CHEBY_1ROW": "1", "PLP_LP_1ROW": "1"})]:
    env = dict(os.environ, PLP_FUZZ_TRIALS=trials, PLP_FUZZ_SEED=seed, **extra)
    t0 = time.time()
    rc = subprocess.call([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-k", "fuzz"],
                         env=env, cwd=ROOT, stdout=subprocess.DEVNULL)
    print("fuzz seed", seed, extra, "rc", rc, "%.1f s" % (time.time() - t0), flush=True)
    bad += rc != 0

import numpy as np  # noqa: E402
from scipy.spatial import ConvexHull  # noqa: E402
from polytope_amd import solvers  # noqa: E402
from polytope_amd.quickhull import quickhull  # noqa: E402
solvers.default_solver = "hip"  # the engine is opt-in
rng = np.random.default_rng(2024)
nb = 0
for trial in range(120):
    d = int(rng.integers(2, 6))
    N = int(rng.integers(d + 2, 20000 if d < 5 else 800))
    P = rng.standard_normal((N, d)) if trial % 2 else rng.random((N, d))
    if trial % 5 == 0:
        P[N // 2:] = P[:N - N // 2]
    np.random.seed(trial)
    A, b, V = quickhull(P)
    ref = np.unique(P[np.unique(ConvexHull(P).vertices)], axis=0)
    Vu = np.unique(V, axis=0)
    ok = Vu.shape == ref.shape and np.array_equal(Vu[np.lexsort(Vu.T[::-1])], ref[np.lexsort(ref.T[::-1])]) \
        and np.max(A @ P.T - b[:, None]) < 1e-7
    nb += not ok
    if not ok:
        print("hull mismatch", trial, d, N, flush=True)
print("quickhull vs scipy ConvexHull: 120 inputs, mismatches", nb)
print("SOAK", "FAILED" if (bad or nb) else "OK")
