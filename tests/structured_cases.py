"""Seeded STRUCTURED (16, 3) polytopes for reduce() -- everything random data does not produce: exact ties in the ratio
tests, duplicated and shifted-parallel rows, rows through a common vertex, tangent (weakly redundant) rows, rows whose
slack sits within a few abs_tol of the keep threshold, corner cuts that are redundant only because of each other, and
integer-lattice normals.  The F2 presolve of the HIP build settles rows ahead of the reference's order and applies the
reference's in-place `h[k] += 0.1 ... -= 0.1` round trip (polytope/polytope.py:1149-1151) to them at once, so this is the
input on which a reordering could show (shared by the CPU oracle tests and the GPU parity tests)."""
import itertools

import numpy as np

M, D = 16, 3
_BOX = np.vstack([np.eye(D), -np.eye(D)])
_LATTICE = np.array([v for v in itertools.product((-1, 0, 1), repeat=D) if any(v)], dtype=float)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def family_names():
    return ["grid_dup", "vertex_fan", "tangent", "near_tol", "corner_cuts", "lattice", "ulp_twins", "mixed"]


def structured_polytopes(B, seed=11):
    """-> A[B, 16, 3] (unit rows unless stated), b[B, 16], family[B] (index into family_names())."""
    rng = np.random.default_rng(seed)
    A = np.zeros((B, M, D))
    b = np.zeros((B, M))
    fam = np.arange(B) % len(family_names())
    for k in range(B):
        f = fam[k]
        lo = np.round(rng.integers(-3, 3, D) * 0.5, 1)           # a grid cell [lo, lo + w]
        w = rng.choice([0.5, 1.0, 2.0], D)
        hi = lo + w
        box_b = np.r_[hi, -lo]
        rows, rhs = [_BOX], [box_b]
        if f == 0:    # grid_dup: box rows again -- same plane, shifted outwards by 0.1 / 0.5, shifted by one abs_tol
            idx = rng.integers(0, 6, 10)
            shift = rng.choice([0.0, 0.0, 0.1, 0.5, 1e-7, -1e-7, 2e-7], 10)
            rows.append(_BOX[idx]); rhs.append(box_b[idx] + shift)
        elif f == 1:  # vertex_fan: ten planes through the corner `hi`, normals in the positive orthant (supporting)
            n = _unit(np.abs(rng.standard_normal((10, D))) + 0.05)
            rows.append(n); rhs.append(n @ hi)
        elif f == 2:  # tangent: planes touching the box in a vertex, an edge or a facet (support value exactly)
            n = _unit(rng.choice([-1.0, 0.0, 1.0, 0.5], (10, D)) + np.r_[1e-3, 0, 0])
            sup = np.where(n > 0, n * hi, n * lo).sum(1)
            rows.append(n); rhs.append(sup)
        elif f == 3:  # near_tol: supporting planes pulled in / pushed out by 0 .. 3 abs_tol
            n = _unit(rng.standard_normal((10, D)))
            sup = np.where(n > 0, n * hi, n * lo).sum(1)
            off = rng.choice([-3e-7, -2e-7, -1.5e-7, -0.5e-7, 0.0, 0.5e-7, 1.5e-7, 3e-7], 10)
            rows.append(n); rhs.append(sup + off)
        elif f == 4:  # corner_cuts: three cuts of the corner `hi`, each 1.5 abs_tol deep, tilted 1e-3 rad apart
            n = _unit(np.ones(D) / np.sqrt(D) + 1e-3 * rng.standard_normal((3, D)))
            rows.append(n); rhs.append(n @ hi - 1.5e-7)
            n2 = _unit(rng.standard_normal((7, D)))
            rows.append(n2); rhs.append(np.where(n2 > 0, n2 * hi, n2 * lo).sum(1) - rng.random(7) * 0.3 * w.min())
        elif f == 5:  # lattice: normals from {-1, 0, 1}^3 with half-integer right-hand sides (rows NOT normalised)
            idx = rng.choice(len(_LATTICE), 10, replace=False)
            n = _LATTICE[idx]
            rows.append(n); rhs.append(np.round((n @ ((lo + hi) / 2)) * 2) / 2 + rng.integers(0, 3, 10) * 0.5)
        elif f == 6:  # ulp_twins: cutting planes twice, the twin's b one ulp up / down / equal
            n = _unit(rng.standard_normal((5, D)))
            bb = np.where(n > 0, n * hi, n * lo).sum(1) - rng.random(5) * 0.4 * w.min()
            twin = np.array([np.nextafter(v, v + s) if s else v for v, s in zip(bb, rng.choice([-1.0, 0.0, 1.0], 5))])
            rows.append(np.vstack([n, n])); rhs.append(np.r_[bb, twin])
        else:         # mixed: a bit of everything around a random cell
            n = _unit(rng.standard_normal((6, D)))
            sup = np.where(n > 0, n * hi, n * lo).sum(1)
            rows.append(n); rhs.append(sup - rng.choice([0.0, 3e-7, 0.2, -0.1], 6) * w.min())
            idx = rng.integers(0, 6, 4)
            rows.append(_BOX[idx]); rhs.append(box_b[idx] + rng.choice([0.0, 0.1], 4))
        R = np.vstack(rows)[:M]
        h = np.concatenate(rhs)[:M]
        perm = rng.permutation(M) if k % 3 else np.arange(M)     # row order matters to the dedupe and to Bland's rule
        A[k], b[k] = R[perm], h[perm]
    return A, b, fam
