"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d).

numpy's Philox is a counter-based generator: (seed, stream) fully determines the data, so
every rank / the CPU baseline / the GPU path regenerate identical inputs without
communication.  `stream` selects a disjoint sub-stream (e.g. the rank).
"""
import numpy as np


def _rng(seed, stream=0):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def random_hpolytopes(B, m, d, seed=0, stream=0, bounded=True):
    """B random H-polytopes, A[B,m,d] unit rows, b[B,m].

    Rows are tangent to spheres of radius 1..2 around the origin (so the origin is strictly
    inside, Chebyshev radius >= 1); with bounded=True the first 2d rows are the box
    |x_i| <= 3, which guarantees boundedness (B=100000, m=16, d=3).
    """
    rng = _rng(seed, stream)
    A = rng.standard_normal((B, m, d))
    A /= np.sqrt(np.sum(A * A, axis=2, keepdims=True))
    b = 1.0 + rng.random((B, m))
    if bounded and m >= 2 * d:
        A[:, :2 * d, :] = np.vstack([np.eye(d), -np.eye(d)])[None]
        b[:, :2 * d] = 3.0
    return np.ascontiguousarray(A), np.ascontiguousarray(b)


def containment_workload(P, N, d=6, m=16, seed=0, stream=0):
    """Config C3: P polytopes (2d box rows |x_i - c_i| <= 1 plus m-2d random tangent rows
    with b in [0.5,1) about the centre c ~ U[-1,1]^d) and N points ~ U[-2.25,2.25]^d as
    column vectors X[d,N]."""
    rng = _rng(seed, stream)
    cen = rng.uniform(-1.0, 1.0, (P, d))
    A = np.zeros((P, m, d))
    b = np.zeros((P, m))
    box = np.vstack([np.eye(d), -np.eye(d)])
    A[:, :2 * d, :] = box[None]
    b[:, :2 * d] = 1.0 + np.einsum("ik,pk->pi", box, cen)
    if m > 2 * d:
        R = rng.standard_normal((P, m - 2 * d, d))
        R /= np.sqrt(np.sum(R * R, axis=2, keepdims=True))
        A[:, 2 * d:, :] = R
        b[:, 2 * d:] = rng.uniform(0.5, 1.0, (P, m - 2 * d)) + np.einsum("pik,pk->pi", R, cen)
    X = rng.uniform(-2.25, 2.25, (d, N))
    return np.ascontiguousarray(A), np.ascontiguousarray(b), np.ascontiguousarray(X)


def quickhull_workload(N, d=8, F=9, seed=0, stream=0):
    """Config C5: N points U[0,1)^d (rows) and F facets.  The first d+1 facets are those of a
    random start simplex (centred at the origin, outward unit normals); further facets are
    random tangent planes at distance 0.2..0.5 -- the points are translated like quickhull
    does (quickhull.py:188-192)."""
    rng = _rng(seed, stream)
    X = rng.random((N, d))
    S = rng.random((d + 1, d))
    xc = S.mean(axis=0)
    S0 = S - xc
    normals, offsets = [], []
    for i in range(d + 1):
        V = np.delete(S0, i, axis=0)
        # hyperplane through the d vertices: solve [V 1][n; -t] = 0 via SVD null-space
        M = np.hstack([V, np.ones((d, 1))])
        _, _, vt = np.linalg.svd(M)
        w = vt[-1]
        n, t = w[:d], -w[d]
        nn = np.linalg.norm(n)
        n, t = n / nn, t / nn
        if n @ V[0] < 0:
            n, t = -n, -t
        normals.append(n)
        offsets.append(t)
    for _ in range(max(0, F - (d + 1))):
        n = rng.standard_normal(d)
        n /= np.linalg.norm(n)
        normals.append(n)
        offsets.append(rng.uniform(0.2, 0.5))
    normals = np.array(normals[:F])
    offsets = np.array(offsets[:F])
    return np.ascontiguousarray(X - xc), np.ascontiguousarray(normals), np.ascontiguousarray(offsets)
