"""Seeded generators of degenerate LPs (shared by the CPU and GPU tests): many constraints through
one vertex, duplicated faces, touching boxes inflated by 1e-7 (is_adjacent, polytope.py:1860-1866),
single-point feasible sets and barely infeasible sets (phase 1)."""
import numpy as np


def degenerate_lps(seed=3, dims=(2, 3, 4, 6, 8), reps=12):
    rng = np.random.default_rng(seed)
    out = []
    for d in dims:
        for _ in range(reps):
            k = 3 * d
            R = -np.abs(rng.standard_normal((k, d)))
            R /= np.linalg.norm(R, axis=1)[:, None]
            G = np.vstack([-np.eye(d), np.ones((1, d)) / np.sqrt(d), R, R[:d]])[:64]
            h = np.r_[np.zeros(d), 1.0, np.zeros(k), np.zeros(d)][:64]
            c = np.abs(rng.standard_normal(d))
            out.append(("vertex", c, G, h))
            out.append(("vertex_far", -c, G, h))
            if 4 * d <= 64:
                I = np.vstack([np.eye(d), -np.eye(d), np.eye(d), -np.eye(d)])
                hb = np.r_[np.ones(d), np.zeros(d), np.ones(d), np.zeros(d)]
                cc = np.r_[np.zeros(d), -1.0]
                out.append(("cube_dup_F1", cc, np.c_[I, np.ones(4 * d)], hb))
                lo2 = np.zeros(d)
                lo2[0] = 1.0
                b2 = np.r_[np.ones(d), np.zeros(d), lo2 + 1, -lo2] + 1e-7
                out.append(("adjacent_F1", cc, np.c_[I, np.ones(4 * d)], b2))
            x0 = rng.standard_normal(d)
            Gp = np.vstack([np.eye(d), -np.eye(d)])
            out.append(("point", rng.standard_normal(d), Gp, np.r_[x0, -x0]))
            out.append(("infeasible_1e-6", rng.standard_normal(d), Gp, np.r_[x0, -x0 - 1e-6]))
    return out
