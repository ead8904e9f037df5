"""GPU parity tests (run with `pytest -m gpu` on a MI355X): the HIP path, called through the
C ABI (include/plp.h), against
  * the golden vectors generated from the imported reference (tests/golden/*.npz),
  * the CPU oracle (oracle/plp_oracle.c) on seeded inputs,
  * scipy.optimize.linprog called exactly as polytope/solvers.py:152-154 does,
and, at the full BASELINE sizes, size-independent properties.

Bars: status / keep masks / booleans / indices exact; Chebyshev radii and objective values
within 1e-9 (north_star); centres only validated as feasible (not unique).
"""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def pa():
    import polytope_amd as pa
    from polytope_amd import _lib
    assert _lib.available(), "libplp_hip.so did not load or no gfx950 device: the HIP path is mandatory"
    return pa


# ------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("gs", [8, 16, 32, 64])
def test_group_primitives(pa, gs):
    from polytope_amd.batch import selftest
    od, ou = selftest(gs)
    lanes = np.arange(64)
    d = ((lanes * 37) % 64 - 20) * 0.5
    u = (lanes * 29) % 61
    for l in range(64):
        base = l - l % gs
        grp = slice(base, base + gs)
        assert od[l] == d[grp].min(), (gs, l)
        assert ou[l] == u[grp].min(), (gs, l)
        assert od[64 + l] == d[base + (l * 7) % gs], (gs, l)
        bal = sum(1 << i for i in range(min(gs, 32)) if (base + i) % 3 == 0)
        assert ou[64 + l] == bal, (gs, l)


# ------------------------------------------------------------------------------ LDS-resident engine (m > 64)
def test_lds_engine_equals_register_engines(pa, monkeypatch):
    """csrc/plp_lds.hip keeps the dictionary in LDS (one LP per wavefront) and applies the pivot rules and the
    arithmetic of the one-row-per-lane engine entry by entry: on LPs both can hold it must return bitwise the same
    numbers and iteration counts (PLP_LDS=1 sends every batch to it)."""
    from degenerate_cases import degenerate_lps
    rng = np.random.default_rng(21)
    for (m, n, B) in [(16, 3, 300), (33, 5, 100), (64, 8, 60), (64, 17, 30), (7, 4, 50)]:
        G = rng.standard_normal((B, m, n))
        G /= np.linalg.norm(G, axis=2, keepdims=True)
        h = rng.random((B, m)) + 0.2
        h[::3] -= 0.6 * rng.random((len(h[::3]), m))   # phase 1
        h[5::11] -= 3.0                                 # infeasible ones
        if m >= 2 * n:
            G[::2, :2 * n] = np.vstack([np.eye(n), -np.eye(n)])[None]
            h[::2, :2 * n] = 3.0
        G[1::7, 0] = 0.0
        c = rng.standard_normal((B, n))
        mrows = rng.integers(max(1, m - 3), m + 1, B).astype(np.int32)
        monkeypatch.setenv("PLP_LP_1ROW", "1")
        ref = pa.lpsolve_batch(c, G, h, m=mrows)
        monkeypatch.delenv("PLP_LP_1ROW")
        monkeypatch.setenv("PLP_LDS", "1")
        got = pa.lpsolve_batch(c, G, h, m=mrows)
        monkeypatch.delenv("PLP_LDS")
        assert np.array_equal(got["status"], ref["status"]) and {0, 2, 3} >= set(np.unique(got["status"]))
        assert np.array_equal(got["iters"], ref["iters"])
        assert np.array_equal(got["fun"], ref["fun"], equal_nan=True) and np.array_equal(got["x"], ref["x"], equal_nan=True)
        # Chebyshev LPs (forced first pivot)
        A = G[:, :, : n - 1] if n > 1 else G
        monkeypatch.setenv("PLP_CHEBY_1ROW", "1")
        ref = pa.cheby_ball_batch(A, h, m=mrows)
        monkeypatch.delenv("PLP_CHEBY_1ROW")
        monkeypatch.setenv("PLP_LDS", "1")
        got = pa.cheby_ball_batch(A, h, m=mrows)
        monkeypatch.delenv("PLP_LDS")
        assert np.array_equal(got["status"], ref["status"])
        assert np.array_equal(got["r"], ref["r"], equal_nan=True) and np.array_equal(got["xc"], ref["xc"], equal_nan=True)
    # degenerate dictionaries (Bland's rule, phase 1 ending with t basic at 0)
    for kind, c, G, h in degenerate_lps():
        monkeypatch.setenv("PLP_LP_1ROW", "1")
        ref = pa.lpsolve_batch(c[None], G[None], h[None])
        monkeypatch.delenv("PLP_LP_1ROW")
        monkeypatch.setenv("PLP_LDS", "1")
        got = pa.lpsolve_batch(c[None], G[None], h[None])
        monkeypatch.delenv("PLP_LDS")
        assert got["status"][0] == ref["status"][0] and got["iters"][0] == ref["iters"][0], kind
        assert np.array_equal(got["fun"], ref["fun"], equal_nan=True), kind


def test_lps_beyond_64_rows(pa, oracle):
    """region_diff stacks m_poly + sum(active rows) (polytope.py:2212-2224; 73 rows at BASELINE config 4): LPs and
    Chebyshev balls with 65..250 rows against the oracle -- random, stacked duplicates, infeasible, unbounded."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(65)
    for (m, n, B) in [(65, 5, 40), (73, 5, 60), (100, 3, 30), (130, 9, 16), (250, 17, 6), (200, 2, 20)]:
        G = rng.standard_normal((B, m, n))
        G /= np.linalg.norm(G, axis=2, keepdims=True)
        x0 = rng.standard_normal((B, n))
        h = np.einsum("bij,bj->bi", G, x0) + rng.random((B, m)) * rng.choice([1.0, 1.0, -0.02], (B, 1))
        G[::2, :2 * n] = np.vstack([np.eye(n), -np.eye(n)])[None]
        h[::2, :2 * n] = 4.0
        G[3::4, m // 2:] = G[3::4, : m - m // 2]          # every fourth LP: the second half repeats the first
        h[3::4, m // 2:] = h[3::4, : m - m // 2]
        c = rng.standard_normal((B, n))
        mrows = rng.integers(65, m + 1, B).astype(np.int32)
        res = pa.lpsolve_batch(c, G, h, m=mrows)
        assert set(np.unique(res["status"])) <= {0, 2, 3}
        for k in range(B):
            mk = mrows[k]
            so, xo, fo, ito = oracle.lp_solve(c[k], G[k, :mk], h[k, :mk])
            assert res["status"][k] == so, (m, n, k, res["status"][k], so)
            if so == 0:
                assert abs(res["fun"][k] - fo) <= TOL * max(1.0, abs(fo)), (m, n, k)
                assert np.max(G[k, :mk] @ res["x"][k] - h[k, :mk]) <= 1e-7
            if k < 6:
                sp = linprog(c[k], G[k, :mk], h[k, :mk], None, None, bounds=(None, None), options={"presolve": False})
                assert sp.status == res["status"][k], (m, n, k)
                if sp.status == 0:
                    assert abs(sp.fun - res["fun"][k]) <= TOL * max(1.0, abs(sp.fun))
        if n <= 16:
            ch = pa.cheby_ball_batch(G, h, m=mrows)
            for k in range(B):
                so, ro, _ = oracle.cheby(G[k, :mrows[k]], h[k, :mrows[k]])
                assert ch["status"][k] == so, (m, n, k)
                if so == 0:
                    assert abs(ch["r"][k] - ro) <= TOL * max(1.0, abs(ro))
                    nrm = np.linalg.norm(G[k, :mrows[k]], axis=1)
                    assert np.max(G[k, :mrows[k]] @ ch["xc"][k] + nrm * ch["r"][k] - h[k, :mrows[k]]) <= 1e-8


# ------------------------------------------------------------------------------ raw LPs
def _g1_groups():
    g = load_golden("g1_lp.npz")
    groups = {}
    for i in range(len(g["m"])):
        groups.setdefault((int(g["m"][i]), int(g["n"][i])), []).append(i)
    return g, groups


@pytest.mark.parametrize("route", ["default", "lane groups"])
def test_lp_golden(pa, route, monkeypatch):
    """(`route`: n = 5..16 on the one-LP-per-wavefront kernel -- the default -- or everything on the lane-group kernels.)
    g1: 1152 LPs solved by the reference (scipy.optimize.linprog / HiGHS).  KNOWN DEVIATION, asserted rather than
    hidden: on ONE of them (form F3 at (5,3)) the reference returns status 2 ("infeasible") for an LP that is feasible and unbounded --
    HiGHS's presolve cannot tell the two apart and reports "infeasible"; with presolve off it says 3, which is what a
    simplex finds and what this engine returns.  Every other status is the reference's own.  (Callers: cheby_ball
    treats 2 and 3 alike, reduce's F2 can be neither; bounding_box of an UNBOUNDED polytope is where it shows: the
    reference writes l = 0, u = l on 2 and -inf/+inf on 3, polytope.py:1378-1402.)"""
    if route == "lane groups":
        monkeypatch.setenv("PLP_LP_WIDE", "0")
    g, groups = _g1_groups()
    worst = 0.0
    deviating = np.nonzero(g["status"] != g["status_nopresolve"])[0]
    assert len(deviating) == 1 and np.all(g["status"][deviating] == 2) and np.all(g["status_nopresolve"][deviating] == 3)
    for (m, n), idx in groups.items():
        idx = np.array(idx)
        c = g["c"][idx, :n]
        G = g["G"][idx, :m * n].reshape(-1, m, n)
        h = g["h"][idx, :m]
        res = pa.lpsolve_batch(c, G, h)
        st_ref = np.where(g["status"][idx] == g["status_nopresolve"][idx], g["status"][idx],
                          g["status_nopresolve"][idx])  # HiGHS presolve quirk, see test_oracle_golden
        assert np.array_equal(res["status"], st_ref), (m, n, np.nonzero(res["status"] != st_ref))
        ok = st_ref == 0
        err = np.abs(res["fun"][ok] - g["fun"][idx][ok])
        assert np.all(err <= TOL * np.maximum(1.0, np.abs(g["fun"][idx][ok]))), (m, n, err.max())
        worst = max(worst, err.max() if err.size else 0.0)
        # returned x is feasible and attains fun
        x = res["x"][ok]
        viol = np.einsum("bij,bj->bi", G[ok], x) - h[ok]
        assert viol.max() <= 1e-7
        assert np.all(np.isnan(res["fun"][~ok])) and np.all(np.isnan(res["x"][~ok]))
    print("worst |fun - scipy| over g1:", worst)


def test_lp_vs_oracle_and_scipy(pa, oracle):
    from scipy.optimize import linprog
    rng = np.random.default_rng(5)
    for (m, n, B) in [(16, 3, 256), (16, 4, 256), (8, 2, 128), (40, 7, 96), (64, 17, 48), (3, 5, 64)]:
        G = rng.standard_normal((B, m, n))
        G /= np.linalg.norm(G, axis=2, keepdims=True)
        x0 = rng.standard_normal((B, n))
        h = np.einsum("bij,bj->bi", G, x0) + rng.random((B, m)) * rng.choice([1.0, 1.0, -0.02], (B, 1))
        if m >= 2 * n:
            G[:, :2 * n] = np.vstack([np.eye(n), -np.eye(n)])[None]
            h[:, :2 * n] = 4.0
        c = rng.standard_normal((B, n))
        mrows = rng.integers(max(1, m - 3), m + 1, B).astype(np.int32)
        res = pa.lpsolve_batch(c, G, h, m=mrows)
        for k in range(B):
            mk = mrows[k]
            so, xo, fo, _ = oracle.lp_solve(c[k], G[k, :mk], h[k, :mk])
            assert res["status"][k] == so, (m, n, k, res["status"][k], so)
            if so == 0:
                assert abs(res["fun"][k] - fo) <= TOL * max(1.0, abs(fo)), (m, n, k)
            if k < 24:
                sp = linprog(c[k], G[k, :mk], h[k, :mk], None, None, bounds=(None, None), options={"presolve": False})
                assert sp.status == res["status"][k], (m, n, k, sp.status, res["status"][k])
                if sp.status == 0:
                    assert abs(sp.fun - res["fun"][k]) <= TOL * max(1.0, abs(sp.fun))


def test_lp_kernel_variants(pa, oracle, monkeypatch):
    """lpsolve batches: n = 5..16 go to the one-LP-per-wavefront kernel (lp_w_kernel, both phases); the lane-group route
    (n <= 4, n = 17, PLP_LP_WIDE=0) takes two kernels -- origin-feasible LPs the fast path (four rows per lane for n <= 8,
    two for n = 9..17), the rest (phase 1 needed beyond n = 4, Bland cases) the two-phase kernel in a second launch.  A mixed
    batch must agree LP by LP with the oracle and with the two-phase kernel alone (PLP_LP_1ROW=1) on either route."""
    rng = np.random.default_rng(12)
    for (m, n, B) in [(16, 3, 600), (12, 2, 200), (30, 5, 150), (64, 8, 60), (5, 4, 100), (30, 9, 80), (64, 12, 40),
                      (64, 17, 30), (24, 11, 60)]:
        G = rng.standard_normal((B, m, n))
        G /= np.linalg.norm(G, axis=2, keepdims=True)
        h = rng.random((B, m)) + 0.2
        h[::3] -= 0.6 * rng.random((len(h[::3]), m))        # a third of the LPs: some h_i < 0 -> phase 1
        if m >= 2 * n:
            G[:, :2 * n] = np.vstack([np.eye(n), -np.eye(n)])[None]
            h[:, :2 * n] = 3.0
        G[1::7, 0] = 0.0                                      # zero rows (kept feasible)
        c = rng.standard_normal((B, n))
        mrows = rng.integers(max(1, m - 3), m + 1, B).astype(np.int32)
        res = pa.lpsolve_batch(c, G, h, m=mrows)   # (n = 5..16: one LP per wavefront, both phases in one kernel)
        monkeypatch.setenv("PLP_LP_WIDE", "0")     # the lane-group kernels for every n
        res_lg = pa.lpsolve_batch(c, G, h, m=mrows)
        monkeypatch.delenv("PLP_LP_WIDE")
        monkeypatch.setenv("PLP_LP_1ROW", "1")
        ref = pa.lpsolve_batch(c, G, h, m=mrows)
        monkeypatch.delenv("PLP_LP_1ROW")
        ok = ref["status"] == 0
        for got in (res, res_lg):
            assert np.array_equal(got["status"], ref["status"]) and set(np.unique(got["status"])) <= {0, 2, 3}
            assert np.array_equal(got["iters"], ref["iters"])     # the same vertex path
            assert np.allclose(got["fun"][ok], ref["fun"][ok], rtol=0, atol=1e-12)
        if 5 <= n <= 16:  # same dictionary arithmetic: the one-LP-per-wavefront engine returns the general kernel's bits
            assert np.array_equal(res["x"][ok].view(np.uint64), ref["x"][ok].view(np.uint64)), (m, n)
        for k in range(0, B, 5):
            so, xo, fo, _ = oracle.lp_solve(c[k], G[k, :mrows[k]], h[k, :mrows[k]])
            assert res["status"][k] == so, (m, n, k)
            if so == 0:
                assert abs(res["fun"][k] - fo) <= TOL * max(1.0, abs(fo))
                assert np.max(G[k, :mrows[k]] @ res["x"][k] - h[k, :mrows[k]]) <= 1e-7


def test_lp_one_per_wavefront_edge_cases(pa, oracle, monkeypatch):
    """lp_w_kernel (n = 5..16) on the corners of the contract: no rows, zero rows (vacuous and infeasible), zero cost,
    infeasible and unbounded LPs with and without phase 1, equality-like row pairs (degenerate phase 1: t leaves at 0 or
    stays basic in a redundant row), ragged row counts -- status / x / iterations as the lane-group kernels and the
    oracle return them."""
    rng = np.random.default_rng(8)
    for n in (5, 8, 11, 16):
        m = 2 * n + 6
        B = 12
        G = np.zeros((B, m, n)); h = np.zeros((B, m)); c = rng.standard_normal((B, n)); rows = np.full(B, m, np.int32)
        box = np.vstack([np.eye(n), -np.eye(n)])
        for k in range(B):
            G[k, :2 * n] = box; h[k, :2 * n] = 1.0 + rng.random(2 * n)
            G[k, 2 * n:] = rng.standard_normal((6, n)); h[k, 2 * n:] = 3.0 + rng.random(6)
        rows[0] = 0                                            # no rows: unbounded (status 3) unless c = 0
        rows[1] = 0; c[1] = 0.0                                # no rows, zero cost: optimal at the origin
        G[2, 3] = 0.0; h[2, 3] = -1.0                          # 0 <= -1: infeasible
        G[3, 3] = 0.0; h[3, 3] = 2.0                           # 0 <= 2: vacuous
        h[4, :n] = -2.0; h[4, n:2 * n] = 1.0                   # x <= -2 and -x <= 1: infeasible, found by phase 1
        h[5, :n] = -0.5; h[5, n:2 * n] = 1.5                   # -1.5 <= x <= -0.5: origin infeasible, LP feasible
        G[6, :n] = 0.0; h[6, :n] = 0.0                         # no upper bounds: unbounded for most costs
        h[7, :n] = -1.0; h[7, n:2 * n] = 1.0                   # x = -1 exactly (pairs of opposite rows): degenerate
        G[8, 2 * n] = G[8, 0]; h[8, 2 * n] = h[8, 0]           # duplicated row
        h[9, :2 * n] = 0.0                                      # the box is the single point 0: every pivot degenerate
        c[10] = 0.0                                             # zero cost on a feasible polytope
        rows[11] = n + 2                                        # ragged: fewer rows than the box needs -> unbounded
        got = pa.lpsolve_batch(c, G, h, m=rows)
        monkeypatch.setenv("PLP_LP_WIDE", "0")
        ref = pa.lpsolve_batch(c, G, h, m=rows)
        monkeypatch.delenv("PLP_LP_WIDE")
        assert np.array_equal(got["status"], ref["status"]), (n, got["status"], ref["status"])
        assert np.array_equal(got["iters"], ref["iters"]), (n, got["iters"], ref["iters"])
        ok = ref["status"] == 0
        assert np.allclose(got["x"][ok], ref["x"][ok], rtol=0, atol=1e-12) and np.isnan(got["x"][~ok]).all()
        for k in range(B):
            so, xo, fo, _ = oracle.lp_solve(c[k], G[k, :rows[k]], h[k, :rows[k]])
            assert got["status"][k] == so, (n, k, got["status"][k], so)
            if so == 0:
                assert abs(got["fun"][k] - fo) <= TOL * max(1.0, abs(fo)), (n, k)
        assert list(got["status"][[0, 1, 2, 4, 5, 10]]) == [3, 0, 2, 2, 0, 0]


def test_lp_edge_inputs(pa):
    # known-answer cases of the reference's tests (polytope_test.py:510-548)
    r = pa.lpsolve_batch(np.array([[1.0]]), np.array([[[-1.0]]]), np.array([[1.0]]))
    assert r["status"][0] == 0 and r["x"][0, 0] == -1.0
    r = pa.lpsolve_batch(np.array([[1.0, 1.0]]), np.array([[[-1.0, 0], [0, -1.0]]]), np.array([[1.0, 1.0]]))
    assert r["status"][0] == 0 and np.array_equal(r["x"][0], [-1.0, -1.0])
    # empty batch, zero rows, zero-row constraints, unbounded, infeasible
    r = pa.lpsolve_batch(np.zeros((0, 2)), np.zeros((0, 4, 2)), np.zeros((0, 4)))
    assert r["status"].shape == (0,)
    r = pa.lpsolve_batch(np.array([[1.0]]), np.zeros((1, 0, 1)), np.zeros((1, 0)))
    assert r["status"][0] == 3
    r = pa.lpsolve_batch(np.array([[1.0], [1.0]]), np.array([[[0.0], [-1.0]], [[0.0], [-1.0]]]),
                         np.array([[-1.0, 1.0], [1.0, 1.0]]))
    assert list(r["status"]) == [2, 0] and r["x"][1, 0] == -1.0
    r = pa.lpsolve_batch(np.array([[1.0, 0.0]]), np.array([[[1.0, 0], [-1.0, 0], [0, 1.0], [0, -1.0]]]),
                         np.array([[1.0, -2.0, 1.0, 1.0]]))
    assert r["status"][0] == 2
    with pytest.raises(ValueError):
        pa.lpsolve_batch(np.array([[1.0]]), np.array([[[1.0], [-1.0]]]), np.array([[np.inf, 1.0]]))
    # more than 64 rows: the LDS-resident engine (csrc/plp_lds.hip); beyond what 160 KB of LDS hold: ValueError
    r = pa.lpsolve_batch(np.array([[1.0, 1.0]]), np.tile(np.array([[-1.0, 0], [0, -1.0], [1.0, 1.0]]), (30, 1))[None],
                         np.tile(np.array([1.0, 1.0, 5.0]), 30)[None])
    assert r["status"][0] == 0 and np.array_equal(r["x"][0], [-1.0, -1.0])
    with pytest.raises(ValueError):
        pa.lpsolve_batch(np.zeros((1, 2)), np.zeros((1, 6000, 2)), np.zeros((1, 6000)))


# ------------------------------------------------------------------------------ bounding box
@pytest.mark.parametrize("variant", [None, "PLP_CHEBY_RETRY_ALL", "PLP_BBOX_WIDE"])
def test_bbox_vs_oracle(pa, oracle, variant, monkeypatch):
    """Fused bounding-box kernel (Chebyshev LP, then 2d LPs from its centre) against the oracle's 2d generic LPs
    (bounding_box, polytope.py:1367-1409): boxes within 1e-9, +-inf in the same places; polytopes the kernel hands
    back (status 1) are exactly those without a usable centre."""
    from polytope_amd.synth import random_hpolytopes
    if variant:  # (PLP_BBOX_WIDE=1: d >= 5 on the one-polytope-per-wavefront kernel whatever the batch size)
        monkeypatch.setenv(variant, "1")
    rng = np.random.default_rng(41)
    for (m, d, B) in [(16, 3, 300), (10, 2, 200), (32, 6, 60), (64, 8, 24), (6, 4, 80), (12, 1, 50), (20, 5, 60), (48, 7, 40)]:
        A, b = random_hpolytopes(B, m, d, seed=5 * m + d, bounded=(m >= 2 * d))
        cen = rng.standard_normal((B, d))                     # move them off the origin (generic LPs: phase 1)
        b = b + np.einsum("bij,bj->bi", A, cen)
        for k in range(3, B, 9):
            b[k, 0] = b[k, 0] - 6.0                            # empty or nearly so
        for k in range(5, B, 13):
            A[k, : min(2 * d, m)] = A[k, -1]                   # drop the box: some become unbounded
            b[k, : min(2 * d, m)] = b[k, -1]
        mrows = rng.integers(max(1, m - 4), m + 1, B).astype(np.int32)
        res = pa.bbox_batch(A, b, m=mrows)
        n_done = 0
        for k in range(B):
            mk = mrows[k]
            lb, ub, bad = oracle.bounding_box(A[k, :mk], b[k, :mk])
            r, xc = oracle.cheby_ball(A[k, :mk], b[k, :mk])
            if res["status"][k] == 0:
                n_done += 1
                assert bad == 0 and r >= 1e-6 - 1e-12, (m, d, k, r)
                assert np.allclose(res["lb"][k], lb.ravel(), rtol=0, atol=TOL, equal_nan=False), (m, d, k, res["lb"][k], lb.ravel())
                assert np.allclose(res["ub"][k], ub.ravel(), rtol=0, atol=TOL, equal_nan=False), (m, d, k, res["ub"][k], ub.ravel())
            else:
                assert res["status"][k] == 1 and (r < 1e-6 + 1e-12), (m, d, k, r)
        assert n_done > B // 3 or m < 2 * d, (m, d, n_done)   # (m < 2d: mostly unbounded, handed back)
    # the golden edge cases through the Python layer (kernel + generic LPs for what it hands back)
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    g = load_golden("g3_edge.npz")
    old = solvers.default_solver
    solvers.default_solver = "hip"
    try:
        for name in g["names"]:
            P = pc.Polytope(g[f"{name}_A"], g[f"{name}_b"], normalize=False)
            if P.A.size == 0 or not np.all(np.isfinite(P.b)):
                continue
            lo, hi = pc.bounding_box(P)
            assert np.allclose(lo.ravel(), g[f"{name}_lb"].ravel(), rtol=0, atol=TOL), (name, lo.ravel(), g[f"{name}_lb"])
            assert np.allclose(hi.ravel(), g[f"{name}_ub"].ravel(), rtol=0, atol=TOL), (name, hi.ravel(), g[f"{name}_ub"])
    finally:
        solvers.default_solver = old


@pytest.mark.parametrize("wide", [None, "dense", "lazy"])
def test_bbox_golden(pa, wide, monkeypatch):
    """The reference's own bounding_box outputs (g10: origin outside, unbounded, empty): the fused kernel where it
    answers, and the Python layer (kernel + generic LPs for what it hands back) for every case.  `wide`: the d >= 5
    cases on the one-polytope-per-wavefront kernel (PLP_BBOX_WIDE=1), its 2d LPs with / without a stored dictionary."""
    if wide:
        monkeypatch.setenv("PLP_BBOX_WIDE", "1")
        monkeypatch.setenv("PLP_BBOX_WDENSE", "1" if wide == "dense" else "0")
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    g = load_golden("g10_bbox.npz")
    old = solvers.default_solver
    solvers.default_solver = "hip"
    n_kernel = 0
    try:
        for i in range(len(g["m"])):
            m, d = int(g["m"][i]), int(g["d"][i])
            A, b = g["A"][i, :m * d].reshape(m, d).copy(), g["b"][i, :m].copy()
            lb, ub = g["lb"][i, :d], g["ub"][i, :d]
            res = pa.bbox_batch(A[None], b[None])
            if res["status"][0] == 0:
                n_kernel += 1
                assert np.array_equal(np.isinf(res["lb"][0]), np.isinf(lb)) and np.array_equal(np.isinf(res["ub"][0]), np.isinf(ub)), i
                assert np.allclose(res["lb"][0], lb, rtol=0, atol=TOL) and np.allclose(res["ub"][0], ub, rtol=0, atol=TOL), i
            lo, hi = pc.bounding_box(pc.Polytope(A, b, normalize=False))
            assert np.allclose(lo.ravel(), lb, rtol=0, atol=TOL) and np.allclose(hi.ravel(), ub, rtol=0, atol=TOL), (i, lo.ravel(), lb)
    finally:
        solvers.default_solver = old
    assert n_kernel >= 40


# ------------------------------------------------------------------------------ Chebyshev
def test_cheby_golden_edges(pa, oracle):
    g = load_golden("g3_edge.npz")
    for name in g["names"]:
        A, b = g[f"{name}_A"], g[f"{name}_b"]
        res = pa.cheby_ball_batch(A[None], b[None])
        st, r, xc = int(res["status"][0]), float(res["r"][0]), res["xc"][0]
        rr = r if (st == 0 and r >= 0) else 0.0
        assert abs(rr - float(g[f"{name}_r"])) <= TOL, name
        assert (st == 0) == (int(g[f"{name}_f1status"]) == 0), name
        if st == 0 and r > 0:
            nrm = np.sqrt((A * A).sum(1))
            assert np.max(A @ xc + nrm * r - b) <= 1e-9, name


def test_cheby_vs_oracle(pa, oracle):
    from polytope_amd.synth import random_hpolytopes
    for (m, d, B) in [(16, 3, 512), (10, 2, 256), (32, 6, 128), (64, 16, 64), (7, 5, 128), (24, 9, 64)]:
        A, b = random_hpolytopes(B, m, d, seed=11, bounded=False)
        mrows = np.random.default_rng(3).integers(max(1, m - 4), m + 1, B).astype(np.int32)
        res = pa.cheby_ball_batch(A, b, m=mrows)
        for k in range(B):
            so, ro, xo = oracle.cheby(A[k, :mrows[k]], b[k, :mrows[k]])
            assert res["status"][k] == so, (m, d, k)
            if so == 0:
                assert abs(res["r"][k] - ro) <= TOL, (m, d, k, res["r"][k], ro)
                Ak = A[k, :mrows[k]]
                assert np.max(Ak @ res["xc"][k] + res["r"][k] - b[k, :mrows[k]]) <= 1e-9


def test_cheby_and_reduce_on_rows_a_hair_apart(pa, oracle):
    """tests/golden/twin_rows.npz (see tests/test_oracle_golden.py): the Chebyshev LP through every engine that takes the shape, and
    the fused reduce (whose F1 it is: a wrong ball made one of these polytopes `empty`).
    Round 6: the stand-alone ball is a CERTIFIED answer (plp_verify.hip) and is held to 1e-9 of the oracle's certified one; the
    fixture's radii are HiGHS's at its default tolerances, which on these rows are 1e-9 .. 3e-8 away from it (asserted per case
    below: that is the reference's own accuracy here, not the kernel's).  The fused reduce's F1 is the uncertified engine
    (parity: the oracle's reduce, same engine, 1e-9) and within 1e-7 of the certified ball."""
    import torch
    g = load_golden("twin_rows.npz")
    for name in "abc":
        A, b, highs = g["A_" + name], g["b_" + name], float(g["r_" + name])
        so, want, _ = oracle.cheby(A, b)
        assert so == 0 and abs(want - highs) <= 1e-7, (name, want, highs)   # HiGHS against the certified optimum
        o = oracle.reduce(A, b)
        for B in (1, 300, 20000):         # (the dispatch changes engine with the batch size)
            At = torch.as_tensor(np.broadcast_to(A, (B,) + A.shape).copy()).cuda()
            bt = torch.as_tensor(np.broadcast_to(b, (B,) + b.shape).copy()).cuda()
            ch = pa.cheby_ball_batch(At, bt)
            assert int(ch["status"].abs().max()) == 0, (name, B)
            assert float((ch["r"] - want).abs().max()) <= 1e-9, (name, B, float((ch["r"] - want).abs().max()))
            rd = pa.reduce_batch(At, bt)
            keep = rd["keep"].cpu().numpy().view(np.uint64)
            assert np.all(keep == np.uint64(o["mask"])) and np.all(rd["flags"].cpu().numpy() == o["flags"]), (name, B)
            assert np.all(rd["nlp"].cpu().numpy() == o["nlp"]) and float((rd["r"] - o["r"]).abs().max()) <= 1e-9, (name, B)
            assert float((rd["r"] - want).abs().max()) <= 1e-7, (name, B)


def _g22_cases():
    g = load_golden("g22_bbox_dup.npz")
    for i in range(len(g["m"])):
        m, d = int(g["m"][i]), int(g["d"][i])
        yield i, g, m, d, g["A"][i, :m * d].reshape(m, d).copy(), g["b"][i, :m].copy()


def _sides_equal(a, o, ext, tol):
    fa, fo = np.isfinite(a), np.isfinite(o)
    return np.array_equal(fa, fo) and np.array_equal(a[~fa], o[~fo]) and bool(np.all(np.abs(a[fa] - o[fo]) <= tol * ext))


def test_bbox_and_cheby_on_rows_a_hair_apart_reference_fixture(pa):
    """g22 (tests/golden/make_golden_bbox_dup.py): the REFERENCE's bounding_box / cheby_ball on 640 polytopes with rows a hair
    apart, slivers, elongated and shifted shapes, d = 2..16 -- where round 5's kernels were off by up to 7e-6 and put +-inf on
    finite sides.  Every box the library returns (fused kernels + verifier; what it hands back: the generic LPs + verifier) within
    1e-9 of the box's extent of the oracle's certified box, +-inf in the same places, on ALL cases; within 2e-9 of the reference's
    own box on the cases where the fixture says the reference is within 1e-9 of the optimum (563 of 640; on the rest HiGHS's 1e-7
    feasibility tolerance, or its treatment of extents beyond 1e9, shows -- recorded per case in the fixture, not a loosened
    tolerance for all).  The same for the Chebyshev radius."""
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    old = solvers.default_solver
    solvers.default_solver = "hip"
    n_ref = n_r = 0
    try:
        for i, g, m, d, A, b in _g22_cases():
            ext = float(g["ext"][i])
            ob = np.concatenate([g["ora_lb"][i, :d], g["ora_ub"][i, :d]])
            if int(g["ora_bad"][i]) == 0:
                lo, hi = pc.bounding_box(pc.Polytope(A, b, normalize=False))
                mine = np.concatenate([lo.ravel(), hi.ravel()])
                assert _sides_equal(mine, ob, ext, 1e-9), (i, str(g["fam"][i]), d, m, mine, ob)
                if g["dev_box"][i] <= 1e-9:
                    n_ref += 1
                    assert _sides_equal(mine, np.concatenate([g["ref_lb"][i, :d], g["ref_ub"][i, :d]]), ext, 2e-9), i
            r, _ = pc.cheby_ball(pc.Polytope(A, b, normalize=False))
            assert abs(r - float(g["ora_r"][i])) <= 1e-9 * max(1.0, abs(float(g["ora_r"][i]))), (i, str(g["fam"][i]), r, float(g["ora_r"][i]))
            if g["dev_r"][i] <= 1e-9:
                n_r += 1
                assert abs(r - float(g["ref_r"][i])) <= 2e-9 * max(1.0, abs(float(g["ref_r"][i]))), i
    finally:
        solvers.default_solver = old
    assert n_ref >= 540 and n_r >= 570, (n_ref, n_r)


def test_small_matrix_entries_as_the_references_solver_reads_them_gpu(pa):
    """g24: the reference's lpsolve / bounding_box on LPs with one entry eps on a ladder around 1e-9 and a lever of 1e3 / 1e6 behind
    it (HiGHS takes |entry| <= 1e-9 for zero) -- through the product's public lpsolve(solver='hip'), its batch form, and
    bounding_box on the 'hip' backend: the reference's answer on every one."""
    import torch
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    from test_oracle_golden import _g24_check
    g = load_golden("g24_small_entries.npz")
    old = solvers.default_solver
    solvers.default_solver = "hip"
    try:
        for i in range(len(g["kind"])):
            def lp(c, G, h):
                r = solvers.lpsolve(c, G, h, solver="hip")
                return int(r["status"]), float(r["fun"]) if r["status"] == 0 else float("nan")

            def bb(A, b):
                lo, hi = pc.bounding_box(pc.Polytope(A.copy(), b.copy(), normalize=False))
                return float(lo[0, 0]), float(hi[0, 0])
            _g24_check(i, g, lp, bb)
    finally:
        solvers.default_solver = old
    lev = np.nonzero(g["kind"] == "lever")[0]
    res = pa.lpsolve_batch(torch.as_tensor(g["c"][lev]).cuda(), torch.as_tensor(g["G"][lev]).cuda(), torch.as_tensor(g["h"][lev]).cuda())
    st, fun = res["status"].cpu().numpy(), res["fun"].cpu().numpy()
    assert np.array_equal(st, g["status"][lev])
    assert np.all(np.abs(fun - g["fun"][lev]) <= 1e-9 * np.maximum(1.0, np.abs(g["fun"][lev])))


def test_polytopes_the_soaks_found(pa, oracle):
    """tests/golden/found/*.npz: the polytopes on which a soak campaign of 25 M found the library and the oracle apart (DESIGN.md
    4.8, "what the last soak campaign found"): the fused reduce against the oracle's (keep mask, flags, LP count exact; as single
    polytopes and inside a batch large enough for the lane kernels), the stand-alone ball and box against the certified oracle."""
    import glob
    import torch
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "found", "*.npz")))
    assert len(files) >= 4
    for f in files:
        z = np.load(f)
        A, b = z["A"], z["b"]
        m, d = A.shape
        ref = oracle.reduce(A, b)
        for B in (1, 36000 if d <= 4 else 300):
            At = torch.as_tensor(np.repeat(A[None], B, 0)).cuda()
            bt = torch.as_tensor(np.repeat(b[None], B, 0)).cuda()
            rd = pa.reduce_batch(At, bt)
            keep = rd["keep"].cpu().numpy().view(np.uint64)
            fl = set(rd["flags"].cpu().numpy().tolist())
            if fl == {33} and not (ref["flags"] & 32):
                # handed back (RF_F1OPEN: the kernel's Chebyshev LP did not end where the oracle's did): "not answered here" --
                # the public reduce(), which re-examines such a polytope through the verified LPs, must keep the oracle's rows
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
                import soak_lane as SL
                assert SL.public_reduce_agrees(A, b, ref["mask"]), (f, B)
                continue
            assert set(int(k) for k in keep) == {ref["mask"]}, (f, B, hex(int(keep[0])), hex(ref["mask"]))
            assert fl == {ref["flags"]} and set(rd["nlp"].cpu().numpy().tolist()) == {ref["nlp"]}, (f, B)
        At, bt = torch.as_tensor(A[None]).cuda(), torch.as_tensor(b[None]).cuda()
        ch = pa.cheby_ball_batch(At, bt)
        st, r, _ = oracle.cheby(A, b)
        assert int(ch["status"][0]) == st, (f, int(ch["status"][0]), st)
        if st == 0:
            assert abs(float(ch["r"][0]) - r) <= 1e-9 * max(1.0, abs(r)), (f, float(ch["r"][0]), r)
        bb = pa.bbox_batch(At, bt)
        lo, hi, bad = oracle.bounding_box(A, b)[:3]
        if bb is not None and int(bb["status"][0]) == 0 and bad == 0:
            fin = np.isfinite(np.concatenate([lo, hi]))
            ext = max(1.0, float(np.abs(np.concatenate([lo, hi])[fin]).max())) if fin.any() else 1.0
            assert _sides_equal(np.concatenate([bb["lb"][0].cpu().numpy(), bb["ub"][0].cpu().numpy()]), np.concatenate([lo, hi]), ext, 1e-9), (
                f, bb["lb"][0].cpu().numpy(), lo, bb["ub"][0].cpu().numpy(), hi)


def test_bbox_batches_on_rows_a_hair_apart(pa, oracle):
    """g22 as BATCHES per shape through plp_bbox_batch / plp_cheby_batch (every fused kernel family: lanes, lane groups, one
    polytope per wavefront and its small-batch form, with and without a stored dictionary) against the fixture's oracle values."""
    import torch
    g = load_golden("g22_bbox_dup.npz")
    shapes = sorted(set(zip(g["d"].tolist(), g["m"].tolist())))
    n_box = 0
    for (d, m) in shapes:
        sel = np.nonzero((g["d"] == d) & (g["m"] == m))[0]
        A = np.stack([g["A"][i, :m * d].reshape(m, d) for i in sel])
        b = np.stack([g["b"][i, :m] for i in sel])
        for rep in (1, 40):   # (the dispatch changes kernel with the batch size)
            At, bt = torch.as_tensor(np.tile(A, (rep, 1, 1))).cuda(), torch.as_tensor(np.tile(b, (rep, 1))).cuda()
            bb = pa.bbox_batch(At, bt)
            ch = pa.cheby_ball_batch(At, bt)
            st, lb, ub = bb["status"].cpu().numpy(), bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy()
            cs, cr = ch["status"].cpu().numpy(), ch["r"].cpu().numpy()
            for k in range(len(sel) * rep):
                i = sel[k % len(sel)]
                want_r = float(g["ora_r"][i])
                got_r = cr[k] if (cs[k] == 0 and cr[k] >= 0) else 0.0
                assert abs(got_r - want_r) <= 1e-9 * max(1.0, want_r), (d, m, rep, k, got_r, want_r)
                if st[k] != 0 or int(g["ora_bad"][i]) != 0:
                    continue
                n_box += 1
                ob = np.concatenate([g["ora_lb"][i, :d], g["ora_ub"][i, :d]])
                assert _sides_equal(np.concatenate([lb[k], ub[k]]), ob, float(g["ext"][i]), 1e-9), (d, m, rep, k, lb[k], ub[k], ob)
    assert n_box >= 15000


@pytest.mark.parametrize("variant", ["PLP_CHEBY_1ROW", "PLP_CHEBY_RETRY_ALL"])
def test_cheby_kernel_variants(pa, oracle, variant, monkeypatch):
    """Chebyshev batches: the one-row-per-lane kernel, and the four-rows-per-lane kernel with every LP
    forced through its hand-over from the fast pivot path to the general engine, against the oracle;
    the pair-adjacency kernel under the same hand-over against its default path."""
    from polytope_amd.synth import random_hpolytopes
    monkeypatch.setenv(variant, "1")
    for (m, d, B) in [(16, 3, 400), (10, 2, 100), (32, 6, 64), (64, 8, 32), (7, 5, 50)]:
        A, b = random_hpolytopes(B, m, d, seed=5 * m + d, bounded=False)
        mrows = np.random.default_rng(m).integers(max(1, m - 4), m + 1, B).astype(np.int32)
        res = pa.cheby_ball_batch(A, b, m=mrows)
        for k in range(B):
            so, ro, xo = oracle.cheby(A[k, :mrows[k]], b[k, :mrows[k]])
            assert res["status"][k] == so, (variant, m, d, k)
            if so == 0:
                assert abs(res["r"][k] - ro) <= TOL, (variant, m, d, k)
    rng = np.random.default_rng(4)
    lo = rng.integers(0, 4, (60, 3)).astype(float)
    A = np.tile(np.vstack([np.eye(3), -np.eye(3)]), (60, 1, 1))
    b = np.concatenate([lo + 1.0, -lo], axis=1)
    forced = pa.adjacent_pairs(A, b)
    monkeypatch.delenv(variant)
    assert np.array_equal(forced, pa.adjacent_pairs(A, b))
    touching = np.all(np.abs(lo[:, None, :] - lo[None, :, :]) <= 1.0, axis=2)
    assert np.array_equal(forced.astype(bool), touching)


# ------------------------------------------------------------------------------ reduce
@pytest.mark.parametrize("fixture", ["g2_reduce.npz", "g15_reduce_mid.npz", "g20_reduce_rows32.npz"])
def test_reduce_golden(pa, fixture):
    """reduce() of the reference: g2 (224 polytopes, d = 2..16) and g15 (50 polytopes of 33..64 rows, d = 5..13: the shapes
    that run one polytope per wavefront, reduce_wdense_kernel)."""
    from polytope_amd import _lib
    g = load_golden(fixture)
    for i in range(len(g["m"])):
        m, d = int(g["m"][i]), int(g["d"][i])
        A = g["A"][i, :m * d].reshape(1, m, d)
        b = g["b"][i, :m].reshape(1, m)
        res = pa.reduce_batch(A, b)
        fl = int(res["flags"][0])
        assert bool(fl & _lib.RF_EMPTY) == bool(g["empty"][i]), i
        if g["empty"][i]:
            continue
        mask = pa.keep_to_bool(res["keep"], m)[0]
        assert np.array_equal(mask, g["mask"][i, :m]), (i, m, d, mask, g["mask"][i, :m])
        assert bool(fl & _lib.RF_MINREP) == bool(g["minrep"][i]), i
        assert abs(res["r"][0] - g["r"][i]) <= TOL, i


def test_reduce_vs_oracle_batches(pa, oracle):
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(9)
    for (m, d, B) in [(16, 3, 1024), (16, 3, 37), (10, 2, 256), (24, 4, 192), (32, 6, 96), (64, 16, 12),
                      (64, 8, 40), (17, 5, 70), (33, 7, 30), (48, 9, 20), (1, 1, 16), (2, 1, 33),
                      (8, 3, 200), (40, 3, 64), (5, 4, 40)]:
        A, b = random_hpolytopes(B, m, d, seed=100 + m + d, bounded=True)
        # make some polytopes degenerate: duplicated rows, unbounded, empty
        for k in range(0, B, 7):
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.1])
        for k in range(3, B, 11):
            b[k, 0] = -5.0  # cuts everything away -> empty or tiny
        mrows = rng.integers(max(min(2, m), m - 5), m + 1, B).astype(np.int32)
        res = pa.reduce_batch(A, b, m=mrows)
        masks = pa.keep_to_bool(res["keep"], m)
        nlp_total = 0
        for k in range(B):
            mk = mrows[k]
            o = oracle.reduce(A[k, :mk], b[k, :mk])
            assert int(res["flags"][k]) == o["flags"], (m, d, k, int(res["flags"][k]), o["flags"])
            assert np.array_equal(masks[k, :mk], o["keep"]), (m, d, k, masks[k, :mk], o["keep"])
            assert not masks[k, mk:].any()
            assert abs(res["r"][k] - o["r"]) <= TOL
            assert int(res["nlp"][k]) == o["nlp"], (m, d, k, int(res["nlp"][k]), o["nlp"])
            nlp_total += o["nlp"]
        assert nlp_total >= B


def test_reduce_beyond_64_rows_vs_oracle(pa, oracle):
    """Fused reduce of polytopes with 65..256 rows (reduce_lds_kernel: rows and dictionary in LDS, keep = W words per
    polytope): keep masks, flags and LP counts exactly the oracle's (polytope.py:1053-1163 has no row limit; the
    stacks of Polytope.intersect, :268-275, and the leaves of region_diff pass 64 rows)."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(19)
    for (m, d, B) in [(80, 3, 48), (100, 4, 40), (128, 8, 24), (65, 2, 32), (250, 16, 4), (200, 6, 10), (96, 12, 8)]:
        A, b = random_hpolytopes(B, m, d, seed=300 + m + d, bounded=True)
        for k in range(0, B, 5):   # duplicated rows (dedupe), shifted copies
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.1])
        for k in range(3, B, 9):
            b[k, 0] = -5.0  # cuts everything away -> empty
        mrows = rng.integers(m - 20, m + 1, B).astype(np.int32)
        mrows[0] = m
        res = pa.reduce_batch(A, b, m=mrows)
        assert res["keep"].shape == (B, (m + 63) // 64)
        masks = pa.keep_to_bool(res["keep"], m)
        for k in range(B):
            mk = mrows[k]
            o = oracle.reduce(A[k, :mk], b[k, :mk])
            assert int(res["flags"][k]) == o["flags"], (m, d, k, int(res["flags"][k]), o["flags"])
            assert np.array_equal(masks[k, :mk], o["keep"]), (m, d, k, np.nonzero(masks[k, :mk] != o["keep"]))
            assert not masks[k, mk:].any()
            assert abs(res["r"][k] - o["r"]) <= TOL
            assert int(res["nlp"][k]) == o["nlp"], (m, d, k, int(res["nlp"][k]), o["nlp"])
        # device-resident call: same bits
        import torch
        rd = pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda(), m=torch.as_tensor(mrows).cuda())
        assert np.array_equal(rd["keep"].cpu().numpy().view(np.uint64), res["keep"])
        assert np.array_equal(rd["nlp"].cpu().numpy(), res["nlp"])
    # a polytope that does not fit the LDS of a CU: PLP_EUNSUPPORTED -> ValueError, never a wrong answer
    A, b = random_hpolytopes(1, 1200, 16, seed=1, bounded=True)
    with pytest.raises(ValueError):
        pa.reduce_batch(A, b)


def _corner_cuts(A, b, rng, ncut=3, depth=1.5e-7, tilt=1e-3):
    """Overwrite the last `ncut` rows of every polytope (whose first 2d rows are the box |x_i| <= 3) with planes that
    cut the corner (3, ..., 3) `depth` deep (> abs_tol: each alone is irredundant), tilted against each other by
    `tilt` rad (1 - cos > abs_tol: the dedupe lets them pass).  With the others present each cuts less than abs_tol: the
    reference's F2 LPs, which all run over ALL rows (polytope.py:1142-1160), drop every one of them."""
    B, m, d = A.shape
    v = np.full(d, 3.0)
    for k in range(B):
        for t in range(ncut):
            n = np.ones(d) / np.sqrt(d) + tilt * rng.standard_normal(d)
            n /= np.linalg.norm(n)
            A[k, m - 1 - t] = n
            b[k, m - 1 - t] = n @ v - depth


def test_reduce_mutually_redundant_rows(pa, oracle):
    """Rows that are redundant only because of each other: every F2 LP sees every live row, also the rows an earlier F2
    LP found redundant (ref :1142-1160 solves over G = A_arr throughout) -- on the register kernels (<= 64 rows) and on
    reduce_lds_kernel (> 64 rows), exactly the oracle's masks."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(23)
    dropped_all = 0
    for (m, d, B) in [(16, 3, 64), (24, 4, 32), (40, 6, 16), (70, 3, 24), (90, 4, 16), (130, 5, 8)]:
        A, b = random_hpolytopes(B, m, d, seed=500 + m, bounded=True)
        b[:, 2 * d:] += 2.5   # the random rows stay clear of the corner (3, ..., 3)
        _corner_cuts(A, b, rng)
        res = pa.reduce_batch(A, b)
        masks = pa.keep_to_bool(res["keep"], m)
        for k in range(B):
            o = oracle.reduce(A[k], b[k])
            assert int(res["flags"][k]) == o["flags"], (m, d, k)
            assert np.array_equal(masks[k], o["keep"]), (m, d, k, np.nonzero(masks[k] != o["keep"]))
            assert int(res["nlp"][k]) == o["nlp"], (m, d, k)
            dropped_all += int(not o["keep"][m - 3:].any())
    assert dropped_all > 0   # the case is there: somewhere all three cuts go


def test_reduce_structured_ties_golden_and_bench_shape(pa, oracle):
    """Structured (16, 3) polytopes (tests/structured_cases.py): ties in the ratio tests, duplicated / shifted-parallel
    rows, vertex fans, tangent rows, slacks a few abs_tol from the threshold, mutually redundant corner cuts, lattice
    normals, one-ulp twins.  (i) g17: the reference's masks where its verdict does not hang on HiGHS's tolerance;
    (ii) 24 576 of them as ONE batch -- the bench kernel's launch shape, every tile full, the F2 presolve and its early
    `h[k] +- 0.1` round trip (ref :1149-1151) on non-random data -- mask, flags, LP count and b round trip exactly the
    oracle's."""
    from test_oracle_golden import structured_golden
    from structured_cases import structured_polytopes
    from polytope_amd import _lib
    g, A, b = structured_golden()
    res = pa.reduce_batch(A, b)
    masks = pa.keep_to_bool(res["keep"], 16)
    for k in np.nonzero(g["pinned"])[0]:
        assert bool(int(res["flags"][k]) & _lib.RF_EMPTY) == bool(g["empty"][k]), k
        if not g["empty"][k]:
            assert np.array_equal(masks[k], g["keep"][k]), (k, masks[k], g["keep"][k])
            assert bool(int(res["flags"][k]) & _lib.RF_MINREP) == bool(g["minrep"][k]), k
            assert abs(res["r"][k] - g["r_tight"][k]) <= TOL, k
    B = 24576
    A, b, fam = structured_polytopes(B, seed=29)
    res = pa.reduce_batch(A, b)
    masks = pa.keep_to_bool(res["keep"], 16)
    bad = []
    for k in range(B):
        o = oracle.reduce(A[k], b[k])
        if not (int(res["flags"][k]) == o["flags"] and np.array_equal(masks[k], o["keep"])
                and int(res["nlp"][k]) == o["nlp"] and abs(res["r"][k] - o["r"]) <= TOL):
            bad.append((k, int(fam[k]), int(res["flags"][k]), o["flags"], int(res["nlp"][k]), o["nlp"]))
    assert not bad, (len(bad), bad[:8])
    # the same batch on the device-resident path, twice: bit-repeatable
    import torch
    Ad, bd = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    r1, r2 = pa.reduce_batch(Ad, bd), pa.reduce_batch(Ad, bd)
    for key in ("keep", "flags", "nlp", "r"):
        assert torch.equal(r1[key], r2[key]), key
    assert np.array_equal(r1["keep"].cpu().numpy().view(np.uint64).ravel(), res["keep"].ravel())


def test_reduce_simplex_run_counter(pa):
    """plp_reduce_counters: LPs that ran the simplex in the fused reduce launches since the last reset.  Between B (F1 of
    every polytope) and nlp.sum() (what the reference issues); repeatable; equal to nlp.sum() where no presolve runs
    (the lane-group latency form of small batches solves every LP); the host-pointer and the device-pointer calls count alike."""
    import os
    import torch
    from polytope_amd import batch
    from polytope_amd.synth import random_hpolytopes
    batch.reduce_simplex_runs(reset=True)      # (the first call of the context switches the counting on)
    for (m, d, B) in [(16, 3, 20000), (32, 6, 6000), (64, 8, 3000), (64, 12, 2500)]:
        A, b = random_hpolytopes(B, m, d, seed=700 + m, bounded=True)
        Ad, bd = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        batch.reduce_simplex_runs(reset=True)
        res = pa.reduce_batch(Ad, bd)
        n1 = batch.reduce_simplex_runs(reset=True)
        nlp = int(res["nlp"].sum().item())
        assert B <= n1 < nlp, (m, d, B, n1, nlp)          # random polytopes: the presolve settles most F2 LPs
        pa.reduce_batch(Ad, bd)
        assert batch.reduce_simplex_runs(reset=True) == n1
        pa.reduce_batch(A, b)                              # host path (chunked upload): same launches, same count
        assert batch.reduce_simplex_runs(reset=True) == n1
        small = pa.reduce_batch(Ad[:200], bd[:200])
        ns = batch.reduce_simplex_runs(reset=True)
        # small batches too go through a presolve (d <= 3, up to 16 rows: reduce_lane_kernel's four-polytope tiles;
        # d >= 5: one polytope per workgroup)
        assert 200 <= ns < int(small["nlp"].sum().item())
        if d == 3:    # the lane-group latency form (reduce_split_kernel, A/B switch): every LP on the simplex, no presolve
            os.environ["PLP_REDUCE_SPLIT"] = "1"
            try:
                pa.reduce_batch(Ad[:200], bd[:200])
                assert batch.reduce_simplex_runs(reset=True) == int(small["nlp"].sum().item())
            finally:
                del os.environ["PLP_REDUCE_SPLIT"]
    assert batch.reduce_simplex_runs() == 0


def _pyramids(B, m, d, rng):
    """Polytopes with a highly degenerate vertex: m - d - 1 facets through one apex, a simplex-like base
    below it and duplicated / nearly parallel facets: the redundancy LPs pivot degenerately at the apex
    (Bland's rule territory)."""
    A = np.zeros((B, m, d))
    b = np.zeros((B, m))
    for k in range(B):
        apex = rng.standard_normal(d) * 0.3
        apex[-1] = 1.0 + rng.random()
        ns = m - (d + 1)
        N = rng.standard_normal((ns, d))
        N[:, -1] = np.abs(N[:, -1]) + 0.2           # side facets face upwards
        if k % 3 == 0:
            N[1::2] = N[0::2][: N[1::2].shape[0]] + 1e-5 * rng.standard_normal(N[1::2].shape)  # nearly parallel
        N /= np.linalg.norm(N, axis=1)[:, None]
        A[k, :ns] = N
        b[k, :ns] = N @ apex                          # all through the apex
        G = rng.standard_normal((d + 1, d))
        G[:, -1] = -np.abs(G[:, -1]) - 0.5            # base facets face downwards
        G /= np.linalg.norm(G, axis=1)[:, None]
        A[k, ns:] = G
        b[k, ns:] = 1.0 + rng.random(d + 1)
    return A, b


def test_reduce_degenerate_vertices(pa, oracle):
    """Degenerate vertices (many facets through one point): consecutive zero-length pivots switch
    the engine to Bland's rule; on the default kernel such a polytope is handed from the fast pivot
    path to the general engine (second launch).  Masks / flags / LP counts must equal the oracle's."""
    rng = np.random.default_rng(31)
    for (m, d, B) in [(16, 3, 300), (14, 2, 120), (24, 4, 100), (40, 5, 40), (64, 6, 12)]:
        A, b = _pyramids(B, m, d, rng)
        res = pa.reduce_batch(A, b)
        masks = pa.keep_to_bool(res["keep"], m)
        for k in range(B):
            o = oracle.reduce(A[k], b[k])
            assert int(res["flags"][k]) == o["flags"], (m, d, k, int(res["flags"][k]), o["flags"])
            assert np.array_equal(masks[k], o["keep"]), (m, d, k, masks[k], o["keep"])
            assert abs(res["r"][k] - o["r"]) <= TOL and int(res["nlp"][k]) == o["nlp"]


@pytest.mark.parametrize("variant", [None, "PLP_REDUCE_RETRY_ALL", "PLP_REDUCE_R2=0", "PLP_REDUCE_LAZY=0", "PLP_REDUCE_LAZY=1",
                                     "PLP_REDUCE_R1=1"])
def test_reduce_two_rows_per_lane(pa, oracle, variant, monkeypatch):
    """d = 9..16: the fused reduce on two rows per lane (groups of 16 / 32 lanes; PLP_REDUCE_LAZY=0: also beyond 32 rows),
    on the kernel without a stored dictionary (default beyond 32 rows; PLP_REDUCE_LAZY=1: always), its one-row-per-lane
    arithmetic twin (PLP_REDUCE_R1), the hand-over to the general engine, and the one-row-per-lane kernel of round 1 --
    random, degenerate and ragged polytopes vs the oracle."""
    from polytope_amd.synth import random_hpolytopes
    if variant:
        name, _, val = variant.partition("=")
        monkeypatch.setenv(name, val or "1")
    rng = np.random.default_rng(77)
    for (m, d, B) in [(30, 9, 24), (32, 12, 20), (64, 13, 10), (50, 16, 10), (18, 10, 30), (64, 16, 8)]:
        A, b = random_hpolytopes(B, m, d, seed=3 * m + d, bounded=True)
        for k in range(0, B, 4):
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
        for k in range(2, B, 7):
            b[k, 0] = -4.0
        mrows = rng.integers(max(2, m - 6), m + 1, B).astype(np.int32)
        res = pa.reduce_batch(A, b, m=mrows)
        masks = pa.keep_to_bool(res["keep"], m)
        for k in range(B):
            o = oracle.reduce(A[k, :mrows[k]], b[k, :mrows[k]])
            assert int(res["flags"][k]) == o["flags"], (variant, m, d, k, int(res["flags"][k]), o["flags"])
            assert np.array_equal(masks[k, :mrows[k]], o["keep"]), (variant, m, d, k)
            assert abs(res["r"][k] - o["r"]) <= TOL and int(res["nlp"][k]) == o["nlp"]
    A, b = _pyramids(20, 40, 9, rng)   # degenerate vertices at d = 9
    res = pa.reduce_batch(A, b)
    masks = pa.keep_to_bool(res["keep"], 40)
    for k in range(20):
        o = oracle.reduce(A[k], b[k])
        assert int(res["flags"][k]) == o["flags"] and np.array_equal(masks[k], o["keep"]), (variant, k)
        assert abs(res["r"][k] - o["r"]) <= TOL and int(res["nlp"][k]) == o["nlp"]


@pytest.mark.gpu
def test_bbox_wavefronts_per_polytope_bitwise(pa, oracle, monkeypatch):
    """bbox_wsplit_kernel (small batches at d >= 5: F1 on one wavefront, the 2d box LPs over four) against bbox_lazy_kernel (one
    wavefront per polytope): lb, ub, status bit for bit -- bounded, unbounded, empty, ragged polytopes, with and without a
    stored dictionary; a sample of the boxes against the oracle's generic LPs."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(9)
    for (B, m, d) in [(1, 64, 8), (50, 40, 5), (300, 64, 8), (120, 33, 12), (60, 64, 16), (80, 20, 7), (40, 24, 14)]:
        A, b = random_hpolytopes(B, m, d, seed=13 * m + d, stream=0)
        b = b + np.einsum("bij,bj->bi", A, rng.standard_normal((B, d)))   # origin outside in general
        for k in range(2, B, 9):
            b[k, 0] = -50.0                                                # empty
        for k in range(4, B, 9):
            A[k, : m // 2] *= -1.0                                         # unbounded or empty
        rows = rng.integers(max(d + 2, m - 9), m + 1, B).astype(np.int32)
        for mr in (None, rows):
            outs = []
            for ws in ("0", "1"):
                monkeypatch.setenv("PLP_BBOX_WIDE", "1")
                monkeypatch.setenv("PLP_BBOX_WSPLIT", ws)
                outs.append(pa.bbox_batch(A, b, m=mr))
            monkeypatch.delenv("PLP_BBOX_WIDE", raising=False)
            monkeypatch.delenv("PLP_BBOX_WSPLIT", raising=False)
            for key in ("lb", "ub", "status"):
                assert np.array_equal(np.asarray(outs[0][key]).view(np.uint8), np.asarray(outs[1][key]).view(np.uint8)), ((B, m, d), key)
        res = pa.bbox_batch(A, b)   # the default route
        for k in range(0, B, 7):
            if int(res["status"][k]) != 0:
                continue
            lo, hi, bad = oracle.bounding_box(A[k], b[k])
            if bad:
                continue
            assert np.allclose(res["lb"][k], lo, rtol=0, atol=1e-8, equal_nan=True), ((m, d), k)
            assert np.allclose(res["ub"][k], hi, rtol=0, atol=1e-8, equal_nan=True), ((m, d), k)


@pytest.mark.gpu
def test_reduce_wavefronts_per_polytope_edge_inputs(pa, oracle, monkeypatch):
    """The split kernel on the inputs the pipeline treats specially: polytopes of 0 / 1 / 2 rows, zero rows (feasible and
    infeasible right-hand sides), a NaN or an infinity among the rows, every row a duplicate, unbounded polytopes, an empty
    polytope -- every output bit of the one-wavefront form, flags and masks equal to the oracle's where its inputs are finite."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(3)
    for (B, m, d) in [(40, 40, 5), (40, 64, 8), (30, 33, 12), (24, 20, 7), (24, 64, 15)]:
        A, b = random_hpolytopes(B, m, d, seed=31 * m + d, stream=0)
        rows = np.full(B, m, np.int32)
        rows[0], rows[1], rows[2] = 0, 1, 2
        A[3, 4] = 0.0; b[3, 4] = 1.0            # a zero row that holds
        A[4, 5] = 0.0; b[4, 5] = -1.0           # a zero row that cannot: infeasible
        A[5, :] = A[5, 0]; b[5, :] = b[5, 0]     # every row the same
        A[6, : m // 2] *= -1.0                   # likely unbounded / empty
        b[7, :] = -np.abs(b[7, :]) - 1.0         # empty
        rows[8] = d                              # fewer rows than d + 1: unbounded
        Af, bf = A.copy(), b.copy()
        A[9, 3, 1] = np.nan
        b[10, 2] = np.inf
        b[11, 2] = -np.inf
        outs = []
        for ws in ("0", "2", "4"):
            monkeypatch.setenv("PLP_REDUCE_WSPLIT", ws)
            outs.append(pa.reduce_batch(__import__("torch").as_tensor(A).cuda(), __import__("torch").as_tensor(b).cuda(),
                                        m=__import__("torch").as_tensor(rows).cuda()))
        monkeypatch.delenv("PLP_REDUCE_WSPLIT", raising=False)
        outs = [{k: v.cpu().numpy() for k, v in o.items()} for o in outs]
        for key in outs[0]:
            for other in outs[1:]:
                assert np.array_equal(outs[0][key].view(np.uint8), other[key].view(np.uint8)), ((B, m, d), key)
        masks = pa.keep_to_bool(outs[1]["keep"], m)
        for k in list(range(9)) + list(range(12, B, 5)):
            if rows[k] == 0:
                continue
            o = oracle.reduce(Af[k, :rows[k]], bf[k, :rows[k]])
            assert int(outs[1]["flags"][k]) == o["flags"], ((m, d), k, int(outs[1]["flags"][k]), o["flags"])
            assert np.array_equal(masks[k, :rows[k]], o["keep"]) and int(outs[1]["nlp"][k]) == o["nlp"], ((m, d), k)


@pytest.mark.gpu
def test_reduce_wavefronts_per_polytope_bitwise(pa, oracle, monkeypatch):
    """reduce_wsplit_kernel<D, NW> -- the polytope-per-workgroup pipeline with the independent LPs of a polytope spread over
    NW = 2 / 4 wavefronts (the default up to 16 000 / 1 500 polytopes where reduce_wdense_kernel used to run: more than 32 rows at
    d = 5..8, d = 9..13) -- must give every output bit of the one-wavefront form: the h[k] +- 0.1 round trip is a rule there,
    the box LPs and the redundancy LPs run in another order and on other wavefronts.  Batches of 1 .. 3 000, ragged,
    duplicated and infeasible rows, pyramids (degenerate vertices: Bland's rule inside the LPs), a sample against the oracle."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(11)

    def three(A, b, m=None):
        out = []
        for ws in ("0", "2", "4"):
            monkeypatch.setenv("PLP_REDUCE_WSPLIT", ws)
            out.append(pa.reduce_batch(A, b, m=m))
        monkeypatch.delenv("PLP_REDUCE_WSPLIT", raising=False)
        out.append(pa.reduce_batch(A, b, m=m))   # the default choice
        return out

    for (B, m, d) in [(1, 64, 8), (7, 40, 5), (300, 64, 8), (300, 48, 6), (200, 64, 12), (200, 57, 13), (3000, 33, 7),
                      (2, 64, 13), (150, 64, 9), (100, 35, 10)]:
        A, b = random_hpolytopes(B, m, d, seed=5 * m + d, stream=0)
        for k in range(0, B, 5):
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
        for k in range(3, B, 11):
            b[k, 0] = -4.0
        rows = rng.integers(max(d + 2, m - 9), m + 1, B).astype(np.int32)
        for mr in (None, rows):
            one, two, four, default = three(A, b, mr)
            for key in one:
                for other in (two, four, default):
                    assert np.array_equal(one[key].view(np.uint8), other[key].view(np.uint8)), ((B, m, d), key, mr is None)
        masks = pa.keep_to_bool(two["keep"], m)
        for k in range(0, B, 41 if B < 1000 else 307):
            o = oracle.reduce(A[k, :rows[k]], b[k, :rows[k]])
            assert int(two["flags"][k]) == o["flags"] and np.array_equal(masks[k, :rows[k]], o["keep"]), (m, d, k)
            assert abs(two["r"][k] - o["r"]) <= TOL and int(two["nlp"][k]) == o["nlp"]
    for (n, m, d) in [(40, 40, 9), (40, 40, 6), (30, 64, 8)]:
        A, b = _pyramids(n, m, d, rng)
        one, two, four, default = three(A, b)
        for key in one:
            for other in (two, four, default):
                assert np.array_equal(one[key].view(np.uint8), other[key].view(np.uint8)), ("pyramids", m, d, key)


@pytest.mark.parametrize("variant", ["PLP_REDUCE_1ROW", "PLP_REDUCE_RETRY_ALL", "PLP_REDUCE_SPLIT=0", "PLP_REDUCE_SPLIT=1",
                                     "PLP_REDUCE_SPLIT=0,PLP_REDUCE_RETRY_ALL=1", "PLP_REDUCE_SPLIT=1,PLP_REDUCE_RETRY_ALL=1"])
def test_reduce_kernel_variants(pa, oracle, variant, monkeypatch):
    """The mappings of the fused reduce (4 rows per lane with several polytopes per wavefront = the batch form,
    PLP_REDUCE_SPLIT=0; one polytope per wavefront with its LPs spread over the lane groups = the latency form, default
    for small batches; 1 row per lane, round 1) must agree with the oracle; the non-default ones are selected by
    environment.  PLP_REDUCE_RETRY_ALL sends every polytope through the second pass (the hand-over used when the fast
    pivot path meets a dictionary that needs Bland's rule)."""
    from polytope_amd.synth import random_hpolytopes
    for item in variant.split(","):
        name, _, val = item.partition("=")
        monkeypatch.setenv(name, val or "1")
    rng = np.random.default_rng(21)
    for (m, d, B) in [(16, 3, 700), (10, 2, 130), (8, 3, 65), (13, 1, 40)]:
        A, b = random_hpolytopes(B, m, d, seed=7 * m + d, bounded=True)
        for k in range(0, B, 5):
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
        for k in range(2, B, 9):
            b[k, 0] = -4.0
        mrows = rng.integers(max(2, m - 4), m + 1, B).astype(np.int32)
        res = pa.reduce_batch(A, b, m=mrows)
        masks = pa.keep_to_bool(res["keep"], m)
        for k in range(B):
            o = oracle.reduce(A[k, :mrows[k]], b[k, :mrows[k]])
            assert int(res["flags"][k]) == o["flags"], (variant, m, d, k)
            assert np.array_equal(masks[k, :mrows[k]], o["keep"]), (variant, m, d, k)
            assert abs(res["r"][k] - o["r"]) <= TOL and int(res["nlp"][k]) == o["nlp"]


def test_reduce_properties_full_config(pa):
    """BASELINE config 2 at full size (100k polytopes, m=16, d=3): size-independent properties.
    idempotence: reducing the reduced polytope keeps every row; the Chebyshev radius is unchanged;
    every dropped row is redundant at the Chebyshev centre (strictly inactive)."""
    from polytope_amd import _lib
    from polytope_amd.synth import random_hpolytopes
    B, m, d = 100000, 16, 3
    A, b = random_hpolytopes(B, m, d, seed=0)
    res = pa.reduce_batch(A, b)
    masks = pa.keep_to_bool(res["keep"], m)
    assert not np.any(res["flags"] & _lib.RF_EMPTY)
    full = res["flags"] == _lib.RF_MINREP
    assert full.mean() > 0.99
    assert np.all(res["r"] >= 1.0 - 1e-12)
    cnt = masks.sum(1)
    assert cnt.min() >= d + 1
    # pack the reduced polytopes (ragged) and reduce again
    A2 = np.zeros_like(A)
    b2 = np.zeros_like(b)
    order = np.argsort(~masks, axis=1, kind="stable")  # kept rows first, original order
    A2 = np.take_along_axis(A, order[:, :, None], axis=1)
    b2 = np.take_along_axis(b, order, axis=1)
    res2 = pa.reduce_batch(A2, b2, m=cnt.astype(np.int32))
    masks2 = pa.keep_to_bool(res2["keep"], m)
    assert np.array_equal(masks2.sum(1), cnt), "reduce is not idempotent"
    assert np.allclose(res2["r"], res["r"], atol=1e-9, rtol=0)
    # LP count bookkeeping: 1 F1 + 2d F3 + one F2 per row that survived dedupe/prefilter
    assert np.all(res["nlp"][full] >= 1 + 2 * d + cnt[full]) and np.all(res["nlp"] <= 1 + 2 * d + m)
    # checksum of checksums, pinned by the oracle on a sample
    print("kept rows total", int(cnt.sum()), "LPs", int(res["nlp"].sum()))


def test_fuzz_random_shapes(pa, oracle):
    """Random (rows, dimension) over the whole envelope m <= 64, d <= 16, ragged row counts:
    lpsolve / cheby / reduce / bounding boxes against the oracle (status, masks, flags exact; values 1e-9)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("PLP_FUZZ_SEED", "2026")))
    for trial in range(int(os.environ.get("PLP_FUZZ_TRIALS", "40"))):   # soak runs: PLP_FUZZ_TRIALS=2000
        d = int(rng.integers(1, 17))
        m = int(rng.integers(1, 65))
        B = int(rng.integers(1, 40))
        A = rng.standard_normal((B, m, d))
        A /= np.linalg.norm(A, axis=2, keepdims=True)
        b = 0.5 + rng.random((B, m))
        if m >= 2 * d and trial % 3:
            A[:, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None]
            b[:, :2 * d] = 2.0
        mrows = rng.integers(max(1, m - 6), m + 1, B).astype(np.int32)
        ch = pa.cheby_ball_batch(A, b, m=mrows)
        rd = pa.reduce_batch(A, b, m=mrows)
        masks = pa.keep_to_bool(rd["keep"], m)
        c = rng.standard_normal((B, d))
        lp = pa.lpsolve_batch(c, A, b, m=mrows)
        bb = pa.bbox_batch(A, b, m=mrows)
        for k in range(B):
            Ak, bk = A[k, :mrows[k]], b[k, :mrows[k]]
            if bb is not None and bb["status"][k] == 0:   # fused boxes vs the generic LPs of the oracle
                lo, hi, bad = oracle.bounding_box(Ak, bk)
                # (1e-9 relative to the size of the box: nearly unbounded polytopes have corners at 1e4)
                assert bad == 0 and np.allclose(bb["lb"][k], lo, rtol=TOL, atol=TOL) and np.allclose(
                    bb["ub"][k], hi, rtol=TOL, atol=TOL), (trial, m, d, k, bb["lb"][k], lo, bb["ub"][k], hi)
            so, ro, _ = oracle.cheby(Ak, bk)
            assert ch["status"][k] == so, (trial, m, d, k)
            if so == 0:
                assert abs(ch["r"][k] - ro) <= TOL, (trial, m, d, k)
            o = oracle.reduce(Ak, bk)
            assert int(rd["flags"][k]) == o["flags"], (trial, m, d, k, int(rd["flags"][k]), o["flags"])
            assert np.array_equal(masks[k, :mrows[k]], o["keep"]), (trial, m, d, k)
            assert int(rd["nlp"][k]) == o["nlp"]
            s2, x2, f2, _ = oracle.lp_solve(c[k], Ak, bk)
            assert lp["status"][k] == s2, (trial, m, d, k)
            if s2 == 0:
                assert abs(lp["fun"][k] - f2) <= TOL * max(1.0, abs(f2)), (trial, m, d, k)


# ------------------------------------------------------------------------------ contains
def test_contains_golden(pa):
    g = load_golden("g4_contains.npz")
    A, b, X = g["A"], g["b"], g["X"]
    for ti, tol in enumerate(g["tols"]):
        out = pa.contains_batch(A, b, X, abs_tol=float(tol), region=False)
        assert np.array_equal(out.astype(bool), g["res"][ti]), tol
        reg = pa.contains_batch(A, b, X, abs_tol=float(tol), region=True)
        assert np.array_equal(reg.astype(bool), g["reg"][ti]), tol
        ob = pa.contains_batch(g["boxA"][None], g["boxb"][None], g["Xb"], abs_tol=float(tol), region=False)
        assert np.array_equal(ob[0].astype(bool), g["boxres"][ti]), tol


def test_contains_vs_oracle(pa, oracle):
    from polytope_amd.synth import containment_workload
    for (P, N, d, m) in [(64, 20000, 6, 16), (7, 1000, 2, 5), (33, 4099, 9, 20), (5, 513, 16, 40), (20, 3000, 1, 2), (9, 2049, 11, 30), (12, 1500, 4, 10)]:
        A, b, X = containment_workload(P, N, d=d, m=m, seed=d)
        mrows = np.random.default_rng(d).integers(max(1, m - 3), m + 1, P).astype(np.int32)
        for tol in (1e-7, 0.0):
            o = oracle.contains(A, b, np.ascontiguousarray(X.T), abs_tol=tol, mrows=mrows)
            out = pa.contains_batch(A, b, X, abs_tol=tol, m=mrows, region=False)
            assert np.array_equal(out, o), (P, N, d, m, tol, int((out != o).sum()))
            oreg = oracle.contains(A, b, np.ascontiguousarray(X.T), abs_tol=tol, mrows=mrows, region=True)
            reg = pa.contains_batch(A, b, X, abs_tol=tol, m=mrows, region=True)
            assert np.array_equal(reg, oreg)
            assert np.array_equal(reg.astype(bool), out.astype(bool).any(axis=0))
    with pytest.raises(ValueError):
        pa.contains_batch(A, b, np.zeros((3, 4)))


def test_reduce_latency_form_bitwise(pa, monkeypatch):
    """Small batches take the latency form of the fused reduce (reduce_split_kernel: one polytope per wavefront, every
    lane group runs F1 / dedupe on the same rows, then the 2d box LPs and the redundancy LPs are spread over the groups;
    the in-place h[k] +- 0.1 round trip becomes a rule).  Same engine per LP: every output bitwise equal to the batch
    form -- d = 1..16, 3..64 rows, ragged, duplicated and infeasible rows, pyramids, golden fixture g2, batch sizes
    around the switch-over."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(9)

    def both(A, b, m=None):
        monkeypatch.setenv("PLP_REDUCE_SPLIT", "0")
        monkeypatch.setenv("PLP_REDUCE_HALF", "0")   # (the batch form on full tiles: the same rows per lane as the latency form)
        batch = pa.reduce_batch(A, b, m=m)
        monkeypatch.setenv("PLP_REDUCE_SPLIT", "1")
        lat = pa.reduce_batch(A, b, m=m)
        monkeypatch.delenv("PLP_REDUCE_SPLIT")
        monkeypatch.delenv("PLP_REDUCE_HALF")
        for key in batch:
            assert np.array_equal(batch[key].view(np.uint8), lat[key].view(np.uint8)), key
        return lat

    for (m, d) in [(16, 3), (12, 4), (16, 2), (10, 1), (3, 2), (32, 6), (24, 5), (20, 8), (64, 8), (40, 7), (16, 8), (33, 3),
                   (32, 12), (20, 9), (24, 16), (30, 13)]:   # (d >= 9 with up to 32 rows: two rows per lane, 4 groups)
        for B in (1, 7, 130, 1100):
            A, b = random_hpolytopes(B, m, d, seed=11 * m + d + B, stream=0)
            for k in range(0, B, 5):
                j = rng.integers(m)
                A[k, (j + 1) % m] = A[k, j]
                b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
            for k in range(3, B, 11):
                b[k, 0] = -4.0
            rows = rng.integers(max(1, m - 5), m + 1, B).astype(np.int32)
            both(A, b)
            both(A, b, rows)
    A, b = _pyramids(60, 16, 3, rng)
    both(A, b)
    A, b = _pyramids(30, 40, 6, rng)
    both(A, b)
    for fixture in ("g2_reduce.npz", "g15_reduce_mid.npz"):
        g = load_golden(fixture)
        for i in range(len(g["m"])):
            m, d = int(g["m"][i]), int(g["d"][i])
            both(g["A"][i, :m * d].reshape(1, m, d), g["b"][i, :m].reshape(1, m))
    A, b = random_hpolytopes(9000, 16, 3, seed=78, stream=0)   # medium batches: half-size tiles only
    monkeypatch.setenv("PLP_REDUCE_SPLIT", "0")
    monkeypatch.setenv("PLP_REDUCE_HALF", "0")
    full = pa.reduce_batch(A, b)
    monkeypatch.setenv("PLP_REDUCE_HALF", "1")
    half = pa.reduce_batch(A, b)
    monkeypatch.delenv("PLP_REDUCE_HALF")
    monkeypatch.delenv("PLP_REDUCE_SPLIT")
    dflt = pa.reduce_batch(A, b)
    for key in full:   # (half-size tiles hold two rows per lane: a near-tie in F1 may move a centre by an ulp, nothing else)
        if key == "xc":
            assert np.allclose(full[key], half[key], rtol=0, atol=1e-12, equal_nan=True)
            assert np.allclose(full[key], dflt[key], rtol=0, atol=1e-12, equal_nan=True)
        else:
            assert np.array_equal(full[key].view(np.uint8), half[key].view(np.uint8)), key
            assert np.array_equal(full[key].view(np.uint8), dflt[key].view(np.uint8)), key
    A, b = random_hpolytopes(5000, 16, 3, seed=77, stream=0)   # default dispatch around the switch-over
    for B in (4095, 4096, 4097):
        monkeypatch.setenv("PLP_REDUCE_SPLIT", "0")
        monkeypatch.setenv("PLP_REDUCE_HALF", "0")
        ref = pa.reduce_batch(A[:B], b[:B])
        monkeypatch.delenv("PLP_REDUCE_SPLIT")
        monkeypatch.delenv("PLP_REDUCE_HALF")
        got = pa.reduce_batch(A[:B], b[:B])
        for key in ref:
            if key == "xc":
                assert np.allclose(ref[key], got[key], rtol=0, atol=1e-12, equal_nan=True), B
            else:
                assert np.array_equal(ref[key].view(np.uint8), got[key].view(np.uint8)), (B, key)


def test_reduce_without_stored_dictionary_bitwise(pa, oracle, monkeypatch):
    """reduce_lazy_kernel (plp_lazy.hpp) evaluates the entering column and the leaving row of every F2 / F3 pivot from
    the polytope's rows and the pivots so far with the operations, in the order, the dense engine applies to its stored
    dictionary: every output must be bit-identical to the one-row-per-lane instance of that engine, and to the
    two-rows-per-lane kernel in everything but the last bit of a Chebyshev centre -- ragged, duplicated, infeasible rows; pyramids (degenerate vertices: Bland hand-over); shapes whose LPs run past
    the four pivots kept in registers into the private-array steps; a sample against the oracle.  reduce_wdense_kernel (the
    same kernel with the F3 / F2 LPs on the dense one-LP-per-wavefront engine, wide::solve_dense -- the default for more
    than 32 rows up to d = 13) must agree with it in every bit as well, d = 5..16, Bland's rule inside the LP included."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(5)

    def four(A, b, m=None):
        out = []
        for env in ({"PLP_REDUCE_LAZY": "0"}, {"PLP_REDUCE_LAZY": "0", "PLP_REDUCE_R1": "1"},
                    {"PLP_REDUCE_LAZY": "1", "PLP_REDUCE_WDENSE": "0"}, {"PLP_REDUCE_LAZY": "1", "PLP_REDUCE_WDENSE": "1"}):
            for k in ("PLP_REDUCE_LAZY", "PLP_REDUCE_R1", "PLP_REDUCE_WDENSE"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out.append(pa.reduce_batch(A, b, m=m))
        for k in ("PLP_REDUCE_LAZY", "PLP_REDUCE_R1", "PLP_REDUCE_WDENSE"):
            monkeypatch.delenv(k, raising=False)
        return out

    def compare(dense, one_row, lazy, wdense, d, tag):
        for key in dense:
            # the two one-polytope-per-wavefront kernels: every output, every bit
            assert np.array_equal(wdense[key].view(np.uint8), lazy[key].view(np.uint8)), (tag, key)
            if d > 8:  # (PLP_REDUCE_R1 exists from d = 9 on)
                assert np.array_equal(one_row[key].view(np.uint8), lazy[key].view(np.uint8)), (tag, key)
            if key != "xc":
                assert np.array_equal(dense[key].view(np.uint8), lazy[key].view(np.uint8)), (tag, key)
        # (two rows per lane compare the two ratios of a lane by cross-multiplication, one row per lane by quotient:
        # a near-tie can send F1 through another vertex order -- seen once in 130 000 polytopes, centre off by one ulp)
        assert np.allclose(dense["xc"], lazy["xc"], rtol=0, atol=1e-12, equal_nan=True), tag

    steps_seen = 0
    for (B, m, d) in [(600, 64, 16), (600, 64, 12), (600, 48, 9), (500, 33, 9), (400, 36, 14), (300, 64, 13), (300, 40, 10),
                      (300, 20, 12), (200, 64, 11), (200, 57, 15), (5000, 64, 8), (400, 64, 8), (400, 40, 6), (6000, 33, 5),
                      (300, 64, 5), (300, 48, 7), (200, 30, 6)]:
        A, b = random_hpolytopes(B, m, d, seed=7 * m + d, stream=0)
        for k in range(0, B, 5):
            j = rng.integers(m)
            A[k, (j + 1) % m] = A[k, j]
            b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
        for k in range(3, B, 11):
            b[k, 0] = -4.0
        rows = rng.integers(max(d + 2, m - 9), m + 1, B).astype(np.int32)
        for mr in (None, rows):
            dense, one_row, lazy, wdense = four(A, b, mr)
            compare(dense, one_row, lazy, wdense, d, (B, m, d))
        masks = pa.keep_to_bool(wdense["keep"], m)
        for k in range(0, B, 37 if B < 1000 else 211):
            o = oracle.reduce(A[k, :rows[k]], b[k, :rows[k]])
            assert int(wdense["flags"][k]) == o["flags"] and np.array_equal(masks[k, :rows[k]], o["keep"]), (m, d, k)
            assert abs(wdense["r"][k] - o["r"]) <= TOL and int(wdense["nlp"][k]) == o["nlp"]
        steps_seen += int((lazy["nlp"] > 1).sum())
    assert steps_seen > 0
    for (n, m, d) in [(40, 40, 9), (40, 40, 6), (30, 64, 8)]:
        A, b = _pyramids(n, m, d, rng)
        dense, one_row, lazy, wdense = four(A, b)
        compare(dense, one_row, lazy, wdense, d, ("pyramids", m, d))


def test_bbox_latency_form_bitwise(pa, monkeypatch):
    """Small batches of fused bounding boxes run one polytope per wavefront with the 2d LPs spread over the lane groups
    (bbox_split_kernel); PLP_BBOX_SPLIT=0 is the batch form.  Same engine per LP: lb / ub / status bitwise equal --
    d = 1..8, ragged rows, empty and unbounded polytopes."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(13)
    for (m, d) in [(16, 3), (12, 4), (10, 1), (5, 2), (32, 6), (24, 5), (64, 8), (40, 7), (16, 8)]:
        for B in (1, 9, 300):
            A, b = random_hpolytopes(B, m, d, seed=3 * m + d + B, stream=0)
            b[2::7, 0] = -5.0
            rows = rng.integers(max(1, m - 5), m + 1, B).astype(np.int32)
            for mr in (None, rows):
                monkeypatch.setenv("PLP_BBOX_SPLIT", "0")
                ref = pa.bbox_batch(A, b, m=mr)
                monkeypatch.setenv("PLP_BBOX_SPLIT", "1")
                got = pa.bbox_batch(A, b, m=mr)
                monkeypatch.delenv("PLP_BBOX_SPLIT")
                for key in ref:
                    assert np.array_equal(ref[key].view(np.uint8), got[key].view(np.uint8)), (m, d, B, key)


@pytest.mark.parametrize("dense", ["0", "1", None])
def test_bbox_large_dimensions(pa, oracle, dense, monkeypatch):
    """Fused bounding boxes, one polytope per wavefront (bbox_lazy_kernel: Chebyshev LP on the one-LP-per-wavefront engine,
    the 2d LPs from its centre on the same dense engine -- PLP_BBOX_WDENSE=1, the default up to d = 13 -- or without a
    stored dictionary -- =0, the default beyond): d = 9..16, and d = 5..8 with more than 32 rows at batch sizes beyond the
    latency form's; boxes of bounded polytopes against the oracle's generic LPs (1e-9), +-inf on the unbounded sides of
    half-open polytopes, status 1 (handed back, NaN) for empty ones; ragged rows."""
    from polytope_amd.synth import random_hpolytopes
    if dense is not None:
        monkeypatch.setenv("PLP_BBOX_WDENSE", dense)
    rng = np.random.default_rng(31)
    for (B, m, d) in [(40, 64, 16), (40, 40, 9), (30, 33, 12), (30, 20, 10), (20, 57, 13), (1100, 64, 8), (1100, 40, 6), (1100, 33, 5)]:
        A, b = random_hpolytopes(B, m, d, seed=m + d, stream=0)
        A[:, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None] if m >= 2 * d else A[:, :2 * d]
        if m >= 2 * d:
            b[:, :2 * d] = 1.5 + rng.random((B, 2 * d))
            A[::4, 0] = A[::4, 1]                  # x_0 <= .. dropped in favour of a duplicate: unbounded below? no, above
            b[::4, 0] = b[::4, 1]
        b[3::9, 5] = -50.0                         # empty
        rows = rng.integers(max(2 * d if m >= 2 * d else d + 2, m - 5), m + 1, B).astype(np.int32)
        res = pa.bbox_batch(A, b, m=rows)
        nok = 0
        for k in (range(B) if B < 100 else range(0, B, 17)):
            lo, hi, bad = oracle.bounding_box(A[k, :rows[k]], b[k, :rows[k]])
            if res["status"][k] == 0:
                nok += 1
                assert bad == 0, (m, d, k)
                assert np.allclose(res["lb"][k], lo, rtol=0, atol=TOL) and np.allclose(res["ub"][k], hi, rtol=0, atol=TOL), (
                    m, d, k, res["lb"][k], lo, res["ub"][k], hi)
            else:
                assert np.isnan(res["lb"][k]).all() and np.isnan(res["ub"][k]).all()
        assert nok >= (B if B < 100 else len(range(0, B, 17))) // 2, (m, d, nok)
        assert (res["status"][3::9] == 1).all()


def test_reduce_host_batch_chunked_upload_equals_one_copy(pa, monkeypatch):
    """Host-pointer batches of 8 MB and more go to the device in chunks staged by a thread pool while the kernels of
    earlier chunks run (plp_stage.hpp); PLP_STAGE=0 is the single-copy path.  Same outputs bit for bit -- ragged row
    counts, a batch size that is no multiple of the chunk or the tile, two shapes, repeated calls on one context."""
    from polytope_amd.synth import random_hpolytopes
    for (B, m, d, seed) in [(40013, 16, 3, 5), (100000, 16, 3, 6), (9001, 64, 16, 7)]:
        A, b = random_hpolytopes(B, m, d, seed=seed, stream=0)
        rows = np.random.default_rng(seed).integers(max(d + 2, m - 5), m + 1, B).astype(np.int32)
        for mr in (None, rows):
            monkeypatch.setenv("PLP_STAGE", "0")
            ref = pa.reduce_batch(A, b, m=mr)
            monkeypatch.delenv("PLP_STAGE")
            for _ in range(2):
                got = pa.reduce_batch(A, b, m=mr)
                for k in ref:
                    assert np.array_equal(np.asarray(ref[k]).view(np.uint8), np.asarray(got[k]).view(np.uint8)), (B, m, d, k)
    # the other host-pointer LP batches take the same route
    A, b = random_hpolytopes(70001, 16, 3, seed=11, stream=0)
    rows = np.random.default_rng(11).integers(5, 17, 70001).astype(np.int32)
    c = np.random.default_rng(12).standard_normal((70001, 3))
    calls = [lambda: pa.cheby_ball_batch(A, b, m=rows), lambda: pa.bbox_batch(A, b, m=rows),
             lambda: pa.lpsolve_batch(c, A, b, m=rows)]
    for call in calls:
        monkeypatch.setenv("PLP_STAGE", "0")
        ref = call()
        monkeypatch.delenv("PLP_STAGE")
        got = call()
        for k in ref:
            assert np.array_equal(np.asarray(ref[k]).view(np.uint8), np.asarray(got[k]).view(np.uint8)), k


def test_host_batches_reject_inf_and_nan_like_linprog(pa, monkeypatch):
    """inf / nan in c, G, h resp. A, b of a host-pointer LP batch: ValueError (what scipy.optimize.linprog raises behind
    solvers.lpsolve, solvers.py:152-154), found by the library (plp_ctx_set_check_finite) -- by the staging threads of a
    large batch, by one pass before the copy otherwise -- wherever the value sits, and the context keeps working."""
    from polytope_amd.synth import random_hpolytopes
    A, b = random_hpolytopes(60000, 16, 3, seed=9, stream=0)
    good = pa.reduce_batch(A, b)
    for stage in ("1", "0"):
        monkeypatch.setenv("PLP_STAGE", stage)
        for (arr, idx, val) in [(A, (0, 0, 0), np.nan), (A, (59999, 15, 2), np.inf), (b, (31234, 7), -np.inf),
                                (b, (12345, 0), np.nan)]:
            old = arr[idx]
            arr[idx] = val
            with pytest.raises(ValueError, match="inf, nan"):
                pa.reduce_batch(A, b)
            arr[idx] = old
        again = pa.reduce_batch(A, b)
        for k in good:
            assert np.array_equal(good[k], again[k]), k
    monkeypatch.delenv("PLP_STAGE")
    As, bs = A[:50].copy(), b[:50].copy()
    bs[3, 2] = np.nan
    for fn in (pa.reduce_batch, pa.cheby_ball_batch, pa.bbox_batch):
        with pytest.raises(ValueError, match="inf, nan"):
            fn(As, bs)
    with pytest.raises(ValueError, match="inf, nan"):
        pa.lpsolve_batch(np.array([[np.nan]]), np.array([[[1.0], [-1.0]]]), np.array([[1.0, 1.0]]))


def test_staged_batches_from_several_threads(pa):
    """Contexts are per thread (include/plp.h: re-entrant across contexts): four threads push large host batches --
    each through its own staging pool, copy stream and arena -- and small ones in between; every result must equal
    the single-threaded one."""
    import threading
    from polytope_amd.synth import random_hpolytopes
    jobs = [random_hpolytopes(30000 + 1000 * k, 16, 3, seed=40 + k, stream=0) for k in range(4)]
    ref = [pa.reduce_batch(A, b) for A, b in jobs]
    small = random_hpolytopes(50, 12, 4, seed=9, stream=0)
    ref_small = pa.reduce_batch(*small)
    errors = []

    def work(k):
        try:
            A, b = jobs[k]
            for _ in range(3):
                got = pa.reduce_batch(A, b)
                for key in got:
                    assert np.array_equal(got[key].view(np.uint8), ref[k][key].view(np.uint8)), (k, key)
                gs = pa.reduce_batch(*small)
                for key in gs:
                    assert np.array_equal(gs[key].view(np.uint8), ref_small[key].view(np.uint8)), (k, key)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_contains_threshold_form_is_the_subtraction(pa, monkeypatch):
    """contains_kernel compares the dot product with a per-row threshold thr = the smallest double at which
    `(s - b) < tol` (polytope.py:217) stops holding, instead of subtracting per point.  One-dimensional polytopes with
    a = 1 make s = x exactly, so points placed on and around every threshold probe the equivalence directly: b over 40
    decades, both signs, zero, infinities and NaN; tol positive, zero, negative and infinite; x = b + tol and its 4
    neighbours on either side, +-0, +-inf, NaN, the largest doubles (overflowing differences).  Must equal numpy's
    evaluation of the reference expression and the kernel with the subtraction left in (PLP_CONTAINS_THR=0)."""
    rng = np.random.default_rng(77)
    mags = 10.0 ** rng.uniform(-20, 20, 120)
    bs = np.concatenate([mags * rng.choice([-1.0, 1.0], mags.size), [0.0, -0.0, 1.0, 0.1, 1e308, -1e308, 5e-324,
                                                                     np.inf, -np.inf, np.nan]])
    P = bs.size
    A = np.ones((P, 1, 1))
    b = bs.reshape(P, 1)
    for tol in (1e-7, 0.0, -1e-7, 1.0, 1e-300, np.inf, -np.inf):
        xs = [0.0, -0.0, np.inf, -np.inf, np.nan, np.finfo(float).max, -np.finfo(float).max]
        with np.errstate(all="ignore"):
            for bi in bs:
                c = bi + tol
                lo = hi = c
                xs.append(c)
                for _ in range(4):
                    lo = np.nextafter(lo, -np.inf)
                    hi = np.nextafter(hi, np.inf)
                    xs += [lo, hi]
                xs += [bi, np.nextafter(bi, np.inf), np.nextafter(bi, -np.inf)]
            X = np.array(xs, dtype=np.float64).reshape(1, -1)
            want = ((X - b) < tol)  # [P, N]: one row per polytope, the reference's expression
        got = pa.contains_batch(A, b, X, abs_tol=tol, region=False)
        assert np.array_equal(got.astype(bool), want), (tol, int((got.astype(bool) != want).sum()))
        monkeypatch.setenv("PLP_CONTAINS_THR", "0")
        plain = pa.contains_batch(A, b, X, abs_tol=tol, region=False)
        monkeypatch.delenv("PLP_CONTAINS_THR")
        assert np.array_equal(plain, got), tol


# ------------------------------------------------------------------------------ quickhull kernels
@pytest.mark.parametrize("d", [2, 3, 5, 8])
def test_assign_golden(pa, d):
    g = load_golden("g6_quickhull.npz")
    res = pa.assign_batch(g[f"d{d}_X"], g[f"d{d}_normals"], g[f"d{d}_offsets"], 1e-7)
    assert np.array_equal(res["facet"], g[f"d{d}_fop"])
    assert np.array_equal(res["dist"], g[f"d{d}_dist"])   # bit for bit (numpy's order of additions at d = 8)
    assert np.array_equal(res["argmax"], g[f"d{d}_argmax"])


@pytest.mark.parametrize("d", [7, 8, 9, 12, 16])
def test_assign_distance_order_golden(pa, d, monkeypatch):
    """g18: the reference's distances (np.sum(n * p) - d0, quickhull.py:117-121) BIT FOR BIT where numpy's sum changes its
    order of additions (eight partial sums from 8 elements on), first-facet assignment and furthest point -- on the
    few-facets kernel with one, two and four points per lane and on the general kernel."""
    g = load_golden("g18_distance_order.npz")
    X, nrm, off = g[f"d{d}_X"], g[f"d{d}_normals"], g[f"d{d}_offsets"]
    for env in ({}, {"PLP_ASSIGN_PPT": "1"}, {"PLP_ASSIGN_PPT": "2"}, {"PLP_ASSIGN_PPT": "4"}, {"PLP_ASSIGN_SMALL": "0"}):
        for k in ("PLP_ASSIGN_PPT", "PLP_ASSIGN_SMALL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        import torch
        for res in (pa.assign_batch(X, nrm, off, 1e-7),
                    {k: v.cpu().numpy() for k, v in pa.assign_batch(torch.as_tensor(X).cuda(), torch.as_tensor(nrm).cuda(),
                                                                    torch.as_tensor(off).cuda(), 1e-7).items()}):
            assert np.array_equal(res["facet"], g[f"d{d}_fop"]), env
            assert np.array_equal(res["dist"], g[f"d{d}_dist"]), env
            assert np.array_equal(res["argmax"], g[f"d{d}_argmax"]), env


def test_assign_vs_oracle(pa, oracle):
    from polytope_amd.synth import quickhull_workload
    for (N, d, F) in [(100000, 8, 9), (5000, 3, 4), (30000, 8, 64), (20000, 4, 600), (1000, 16, 17), (300001, 8, 9),
                      (70001, 8, 33), (65, 8, 9), (1, 8, 2), (40000, 12, 13), (40000, 9, 65)]:
        X, nrm, off = quickhull_workload(N, d=d, F=F, seed=F)
        X[N // 2] = X[N // 3]  # exact tie of distances: the first maximum must win
        fo, do_, am, mx = oracle.assign(X, nrm, off, 1e-7)
        res = pa.assign_batch(X, nrm, off, 1e-7)
        assert np.array_equal(res["facet"], fo), (N, d, F)
        assert np.array_equal(res["dist"], do_), (N, d, F)  # same operation order -> bitwise
        assert np.array_equal(res["argmax"], am), (N, d, F)
        has = am >= 0
        assert np.array_equal(res["maxd"][has], mx[has])


def test_degenerate_lps(pa, oracle):
    """Bland's rule / phase-1 corner cases on the device: same status as the oracle and as
    scipy (presolve off), objective within 1e-9."""
    from scipy.optimize import linprog
    from degenerate_cases import degenerate_lps
    cases = degenerate_lps()
    by_shape = {}
    for k, (tag, c, G, h) in enumerate(cases):
        by_shape.setdefault(G.shape, []).append(k)
    for (m, n), idx in by_shape.items():
        c = np.array([cases[k][1] for k in idx])
        G = np.array([cases[k][2] for k in idx])
        h = np.array([cases[k][3] for k in idx])
        res = pa.lpsolve_batch(c, G, h)
        for j, k in enumerate(idx):
            so, xo, fo, _ = oracle.lp_solve(c[j], G[j], h[j])
            sp = linprog(c[j], G[j], h[j], None, None, bounds=(None, None), options={"presolve": False})
            assert res["status"][j] == so == sp.status, (cases[k][0], m, n, res["status"][j], so, sp.status)
            if so == 0:
                assert abs(res["fun"][j] - sp.fun) <= TOL * max(1.0, abs(sp.fun)), (cases[k][0], m, n)
            assert res["iters"][j] <= 50 * (m + n) + 100


def test_adjacent_pairs_one_pair_per_wavefront(pa, monkeypatch):
    """Pair LPs at d = 5..16 (adjacent_w_kernel: one stacked, inflated pair per wavefront; the only device kernel for
    d >= 9): grids of boxes (adjacent iff neighbours in the grid, faces and corners), random cells with ragged rows against
    the Chebyshev batch of the host-stacked pairs (r > abs_tol / 10 on b + abs_tol, polytope.py:1843-1866), the lane-group
    kernel (d <= 8, PLP_ADJ_WIDE=0), pair ranges and the overlap matrix."""
    import itertools
    from polytope_amd import batch
    rng = np.random.default_rng(44)
    tol = 1e-7

    def by_cheby(A, b, ms, inflate, thresh):
        n, m_max, d = A.shape
        ii, jj = np.tril_indices(n, -1)
        SA = np.zeros((len(ii), 2 * m_max, d)); Sb = np.zeros((len(ii), 2 * m_max)); sm = np.zeros(len(ii), np.int32)
        for k, (i, j) in enumerate(zip(ii, jj)):
            mi, mj = ms[i], ms[j]
            SA[k, :mi] = A[i, :mi]; SA[k, mi:mi + mj] = A[j, :mj]
            Sb[k, :mi] = b[i, :mi] + inflate; Sb[k, mi:mi + mj] = b[j, :mj] + inflate
            sm[k] = mi + mj
        res = pa.cheby_ball_batch(SA, Sb, m=sm)
        yes = (res["status"] == 0) & (res["r"] > thresh)
        out = np.eye(n, dtype=np.uint8)
        out[ii, jj] = yes; out[jj, ii] = yes
        return out

    for shape in [(3, 2, 2, 2, 2), (2, 2, 2, 2, 2, 2), (3, 2, 2, 1, 1, 1, 1, 1, 2), (2, 2, 1, 1, 3, 1, 1, 1, 1, 1, 1, 2)]:
        d = len(shape)
        lo = np.array(list(itertools.product(*[range(k) for k in shape])), dtype=float)
        n = lo.shape[0]
        A = np.tile(np.vstack([np.eye(d), -np.eye(d)]), (n, 1, 1))
        b = np.concatenate([lo + 1.0, -lo], axis=1)
        adj = pa.adjacent_pairs(A, b)
        want = (np.abs(lo[:, None, :] - lo[None, :, :]).max(axis=2) <= 1).astype(np.uint8)
        assert np.array_equal(adj, want), shape
        if d <= 8:
            monkeypatch.setenv("PLP_ADJ_WIDE", "0")
            assert np.array_equal(pa.adjacent_pairs(A, b), want), shape
            monkeypatch.setenv("PLP_ADJ_WIDE", "1")
            assert np.array_equal(pa.adjacent_pairs(A, b), want), shape
            monkeypatch.delenv("PLP_ADJ_WIDE")
        ii, jj = np.tril_indices(n, -1)
        for (p0, p1) in [(0, len(ii)), (5, 77), (len(ii) - 3, len(ii))]:
            assert np.array_equal(batch.adjacent_pairs_range(A, b, p0, p1), want[ii[p0:p1], jj[p0:p1]])
        ov = pa.overlap_pairs(A, b) if hasattr(pa, "overlap_pairs") else batch.overlap_pairs(A, b)
        assert np.array_equal(ov, np.eye(n, dtype=np.uint8)), shape      # boxes of a grid only touch
    n_adj = n_apart = 0
    for (n, m, d) in [(40, 20, 6), (30, 32, 8), (24, 24, 10), (16, 32, 16), (20, 12, 5)]:
        # boxes with random centres and widths (overlapping, touching nowhere, apart) cut by a few more random half-spaces
        cen = rng.uniform(0.0, 2.0, (n, d)); hw = rng.uniform(0.3, 0.9, (n, d))
        A = np.zeros((n, m, d)); b = np.zeros((n, m))
        A[:, :d] = np.eye(d); A[:, d:2 * d] = -np.eye(d)
        b[:, :d] = cen + hw; b[:, d:2 * d] = -(cen - hw)
        extra = rng.standard_normal((n, m - 2 * d, d)); extra /= np.linalg.norm(extra, axis=2, keepdims=True)
        A[:, 2 * d:] = extra
        b[:, 2 * d:] = np.einsum("nij,nj->ni", extra, cen) + rng.uniform(0.2, 1.5, (n, m - 2 * d))
        ms = rng.integers(max(2 * d, m - 4), m + 1, n).astype(np.int32)
        adj = pa.adjacent_pairs(A, b, m=ms)
        assert np.array_equal(adj, by_cheby(A, b, ms, tol, tol / 10)), (n, m, d)
        n_adj += int(adj.sum()) - n
        n_apart += n * n - int(adj.sum())
        if d <= 8:
            monkeypatch.setenv("PLP_ADJ_WIDE", "0")
            assert np.array_equal(pa.adjacent_pairs(A, b, m=ms), adj), (n, m, d)
            monkeypatch.delenv("PLP_ADJ_WIDE")
        ov = batch.overlap_pairs(A, b, m=ms)
        assert np.array_equal(ov, by_cheby(A, b, ms, 0.0, tol)), (n, m, d)
    assert n_adj > 50 and n_apart > 50, (n_adj, n_apart)


def test_adjacent_pairs_range_and_sharded_single(pa):
    """Slices of the pair space (what one rank computes when the O(n^2) loop is split across GPUs)
    agree with the full matrix; the sharded wrappers with world size 1 agree with the plain calls."""
    import itertools
    import os
    from polytope_amd import batch, dist as pdist
    from polytope_amd.quickhull import quickhull
    from polytope_amd import solvers
    lo = np.array(list(itertools.product(range(5), range(4), range(3))), dtype=float)
    n = lo.shape[0]
    A = np.tile(np.vstack([np.eye(3), -np.eye(3)]), (n, 1, 1))
    b = np.concatenate([lo + 1.0, -lo], axis=1)
    full = pa.adjacent_pairs(A, b)
    ii, jj = np.tril_indices(n, -1)
    npairs = n * (n - 1) // 2
    for (p0, p1) in [(0, npairs), (0, 1), (17, 400), (npairs - 5, npairs), (100, 100)]:
        part = batch.adjacent_pairs_range(A, b, p0, p1)
        assert np.array_equal(part, full[ii[p0:p1], jj[p0:p1]])
    with pytest.raises(ValueError):
        batch.adjacent_pairs_range(A, b, 0, npairs + 1)
    assert np.array_equal(pdist.adjacent_pairs_sharded(A, b).cpu().numpy(), full)
    P = np.random.default_rng(8).standard_normal((20000, 4))
    old, solvers.default_solver = solvers.default_solver, "hip"
    try:
        np.random.seed(4)
        A1, b1, V1 = quickhull(P)
        np.random.seed(4)
        A2, b2, V2 = pdist.quickhull_sharded(P)
    finally:
        solvers.default_solver = old
    # (the library's main loop solves the hyperplane systems of inputs of 4096+ points with its own LU, the host facet
    # graph of the sharded path with numpy.linalg.solve: same rows in the same order, to rounding)
    assert A1.shape == A2.shape and np.allclose(A1, A2, rtol=0, atol=1e-12) and np.allclose(b1, b2, rtol=0, atol=1e-12)
    assert np.array_equal(V1, V2)
    os.environ["PLP_QH_LAPACK_BELOW"] = "1000000000"   # dgesv for every size: bit-identical to the host facet graph
    try:
        solvers.default_solver = "hip"
        np.random.seed(4)
        A3, b3, V3 = quickhull(P)
    finally:
        solvers.default_solver = old
        del os.environ["PLP_QH_LAPACK_BELOW"]
    assert np.array_equal(A3, A2) and np.array_equal(b3, b2) and np.array_equal(V3, V2)


def test_contains_full_config(pa, oracle):
    """BASELINE config 3 at full size (1M points x 10k polytopes, d=6, m=16), through size-independent
    properties: the verdict of a point does not depend on which other points share the launch
    (halves == whole), the OR over polytope subsets equals the whole Region, and a random sample of
    points agrees with the oracle on all 10k polytopes."""
    import torch
    from polytope_amd.synth import containment_workload
    P, N, d, m = 10000, 1000000, 6, 16
    A, b, X = containment_workload(P, N, d=d, m=m, seed=0)
    dev = torch.device("cuda:0")
    At, bt, Xt = (torch.as_tensor(v).to(dev) for v in (A, b, X))
    whole = pa.contains_batch(At, bt, Xt, 1e-7).cpu().numpy()
    assert whole.shape == (N,) and 0 < whole.sum() < N
    h = N // 2 + 7
    halves = np.concatenate([pa.contains_batch(At, bt, Xt[:, :h].contiguous(), 1e-7).cpu().numpy(),
                             pa.contains_batch(At, bt, Xt[:, h:].contiguous(), 1e-7).cpu().numpy()])
    assert np.array_equal(whole, halves)
    parts = [pa.contains_batch(At[s], bt[s], Xt, 1e-7).cpu().numpy() for s in (slice(0, 3333), slice(3333, P))]
    assert np.array_equal(whole, parts[0] | parts[1])
    idx = np.random.default_rng(0).choice(N, 1500, replace=False)
    ref = oracle.contains(A, b, np.ascontiguousarray(X[:, idx].T), abs_tol=1e-7, region=True)
    assert np.array_equal(whole[idx], ref)


def test_assign_full_config(pa, oracle):
    """BASELINE config 5 at full size (1M points, d=8; the 9 facets of a start simplex and 512 synthetic
    facets): a sample agrees with the oracle; every facet's reported furthest point is one of its own
    points, attains the maximum of their distances, and is the lowest index doing so."""
    from polytope_amd.synth import quickhull_workload
    N, d = 1000000, 8
    for F in (9, 512):
        X, nrm, off = quickhull_workload(N, d=d, F=F, seed=0)
        res = pa.assign_batch(X, nrm, off, 1e-7)
        fac, dist, am, mx = res["facet"], res["dist"], res["argmax"], res["maxd"]
        idx = np.random.default_rng(F).choice(N, 20000, replace=False)
        fo, do_, _, _ = oracle.assign(X[idx], nrm, off, 1e-7)
        assert np.array_equal(fac[idx], fo) and np.array_equal(dist[idx], do_)
        assert np.all(dist[fac < 0] == 0.0) and np.all(dist[fac >= 0] > 1e-7)
        best = np.full(F, -np.inf)
        np.maximum.at(best, fac[fac >= 0], dist[fac >= 0])
        for f in range(F):
            if am[f] < 0:
                assert not np.any(fac == f)
                continue
            assert fac[am[f]] == f and dist[am[f]] == best[f] == mx[f]
            assert am[f] == np.nonzero((fac == f) & (dist == best[f]))[0][0]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_force_dist_rccl_path(pa, scaling):
    """bench.py's N > 1 code path (process group on RCCL, exchange buffers, coalesced overlapped all-gather) driven
    at world size 1 with --force-dist, in both scaling modes: the line must carry the contract's fields and the
    LP count of the unsharded run."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "9", "--warmup", "2", "--no-cpu-baseline",
           "--no-end-to-end", "--force-dist", "--scaling", scaling, "--batches", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == scaling and line["n_gpus"] == 1 and line["steps"] == 9 and line["unit"] == "LP/s"
    assert line["value"] > 1e9 and line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    # LP count per step: the mean over the two batches in the rotation 0,1,0,1,... of the region's steps
    from polytope_amd.synth import random_hpolytopes
    import torch
    n = []
    for i in range(2):
        A, b = random_hpolytopes(100000, 16, 3, seed=i, stream=0)
        n.append(int(pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda())["nlp"].sum().item()))
    S = line["config"]["timed_steps_per_region"]   # the 9 steps repeated until a region lasts 50 ms
    assert S % 9 == 0 and abs(line["config"]["lps_per_step"] - sum(n[k % 2] for k in range(S)) / S) < 1e-6
    assert line["parity_ok"] is True and line["parity_checked"] == 200000
    if scaling == "weak":   # the same run carries the partitioned batch as `strong` (at world size 1: the whole batch)
        assert line["strong"]["strong_floor"]["shard_polytopes"] == 100000 and line["strong"]["value"] > 1e9


def test_region_diff_resident_lp_server(pa, monkeypatch):
    """The search's LPs on the resident server (one launch per search, batches through a host-mapped mailbox,
    rdiff_server_kernel) against a launch per batch (PLP_RDIFF_SERVER=0): the same pieces, row for row, on the 81-cell
    grid of g12 and on random overlapping boxes in d = 2..4 -- also when the server retires between batches (an idle
    limit of a few polls: every batch finds it gone or going and restarts it), with a single workgroup, and across
    consecutive searches (the mailbox word carries on)."""
    import itertools
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers, batch
    old, solvers.default_solver = solvers.default_solver, "hip"
    try:
        rng = np.random.default_rng(88)
        cases = []
        shape = (3, 3, 3, 3)
        cells = [pcm.box2poly([[i[k] / 3, (i[k] + 1) / 3] for k in range(4)]) for i in itertools.product(*[range(3)] * 4)]
        A = rng.standard_normal((10, 4))
        A /= np.linalg.norm(A, axis=1)[:, None]
        cases.append((pcm.Polytope(A, 0.35 + A @ (0.5 * np.ones(4))), cells))
        for trial in range(9):
            d = 2 + trial % 3
            n = int(rng.integers(4, 14))
            cen, hw = rng.random((n, d)), rng.uniform(0.05, 0.3, (n, d))
            A = rng.standard_normal((3 * d, d))
            A /= np.linalg.norm(A, axis=1)[:, None]
            cases.append((pcm.Polytope(A, 0.3 * (1 + rng.random(3 * d)) + A @ (0.5 * np.ones(d))),
                          [pcm.box2poly(np.c_[c - w, c + w].tolist()) for c, w in zip(cen, hw)]))
        spy = {}
        orig = batch.region_diff_search

        def counted(*a, **k):
            out = orig(*a, **k)
            spy["batches"] = spy.get("batches", 0) + out[1]["batches"]
            return out
        monkeypatch.setattr(batch, "region_diff_search", counted)

        def run(env):
            for k in ("PLP_RDIFF_SERVER", "PLP_RDIFF_SERVER_IDLE", "PLP_RDIFF_SERVER_WGS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            res = []
            for P, cs in cases:
                try:
                    D = pcm.region_diff(P.copy(), pcm.Region([c.copy() for c in cs]))
                    ps = list(D.list_poly) if isinstance(D, pcm.Region) else ([] if D.A.size == 0 else [D])
                    res.append([(q.A.copy(), q.b.copy()) for q in ps])
                except IndexError:
                    res.append("IndexError")
            return res
        want = run({"PLP_RDIFF_SERVER": "0"})
        assert spy["batches"] > 50 and any(isinstance(w, list) and len(w) > 3 for w in want)
        for env in ({}, {"PLP_RDIFF_SERVER_IDLE": "3"}, {"PLP_RDIFF_SERVER_WGS": "1"}, {"PLP_RDIFF_SERVER_WGS": "7",
                                                                                        "PLP_RDIFF_SERVER_IDLE": "40"}):
            got = run(env)
            for k, (g_, w_) in enumerate(zip(got, want)):
                assert type(g_) is type(w_), (env, k)
                if w_ != "IndexError":
                    assert len(g_) == len(w_), (env, k, len(g_), len(w_))
                    for (A0, b0), (A1, b1) in zip(g_, w_):
                        assert np.array_equal(A0, A1) and np.array_equal(b0, b1), (env, k)
    finally:
        solvers.default_solver = old


def test_region_diff_library_search_equals_host_loop(pa, monkeypatch):
    """region_diff's search runs in the library (plp_region_diff_search: LPs gathered on the device by row index,
    cells an ancestor scan found empty not re-solved, one batch per visited node).  On random overlapping boxes --
    where the reference's INDICES arithmetic takes its odd turns -- it must return exactly the pieces of the host
    loop over batched calls (which is pinned to the reference by g5 / g11 / g12), or raise IndexError where that does.
    d = 2..4: lane-group gather kernels; d = 5, 6: one LP per wavefront (cheby_gather_w_kernel), and PLP_RDIFF_WIDE=0
    (lane groups there too) must give the same pieces."""
    import polytope_amd.polytope as pcm
    from polytope_amd import solvers
    old, solvers.default_solver = solvers.default_solver, "hip"
    try:
        rng = np.random.default_rng(77)
        checked = 0
        for trial in range(34):
            d = 2 + trial % 3 if trial < 24 else 5 + trial % 2
            n = int(rng.integers(3, 14 if d <= 4 else 8))
            cen = rng.random((n, d))
            hw = rng.uniform(0.05, 0.3, (n, d))
            cells = [pcm.box2poly(np.c_[c - w, c + w].tolist()) for c, w in zip(cen, hw)]
            A = rng.standard_normal((3 * d, d))
            A /= np.linalg.norm(A, axis=1)[:, None]
            P = pcm.Polytope(A, 0.3 * (1 + rng.random(3 * d)) + A @ (0.5 * np.ones(d)))
            out = []
            for native in ((True, False) if d <= 4 else (True, False, "lane groups")):
                pcm._RDIFF_NATIVE = bool(native)
                if native == "lane groups":
                    monkeypatch.setenv("PLP_RDIFF_WIDE", "0")
                else:
                    monkeypatch.delenv("PLP_RDIFF_WIDE", raising=False)
                try:
                    D = pcm.region_diff(P.copy(), pcm.Region([c.copy() for c in cells]))
                    ps = list(D.list_poly) if isinstance(D, pcm.Region) else ([] if D.A.size == 0 else [D])
                    out.append([(q.A.copy(), q.b.copy()) for q in ps])
                except IndexError:
                    out.append("IndexError")
            monkeypatch.delenv("PLP_RDIFF_WIDE", raising=False)
            for other in out[1:]:
                assert type(out[0]) is type(other), trial
                if out[0] != "IndexError":
                    assert len(out[0]) == len(other), (trial, d, len(out[0]), len(other))
                    for (A0, b0), (A1, b1) in zip(out[0], other):
                        assert A0.shape == A1.shape and np.allclose(A0, A1, atol=1e-12, rtol=0) and np.allclose(b0, b1, atol=1e-12, rtol=0), trial
            checked += out[0] != "IndexError"
        assert checked >= 17
    finally:
        pcm._RDIFF_NATIVE = True
        solvers.default_solver = old


def test_wide_engine_one_lp_per_wavefront(pa, oracle, monkeypatch):
    """csrc/plp_wide.hip (one LP per wavefront, wave-uniform pivot column, reduced costs in LDS) serves Chebyshev
    batches with d >= 9 and more than 32 rows; PLP_CHEBY_WIDE=1 sends every shape with d >= 5 to it.  Against the
    lane-group kernels (same pivot rules: statuses equal, radii to 1e-12) and the oracle, on random, ragged, unbounded,
    empty, zero-row and degenerate (duplicated cube faces) inputs."""
    rng = np.random.default_rng(164)
    for (B, m, d) in [(300, 64, 16), (200, 64, 12), (200, 40, 9), (200, 64, 7), (200, 20, 5), (64, 1, 6), (100, 17, 16)]:
        A = rng.standard_normal((B, m, d))
        A /= np.linalg.norm(A, axis=2, keepdims=True)
        b = 1.0 + rng.random((B, m))
        if m >= 2 * d:
            A[::2, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None]      # bounded ones
            b[::2, :2 * d] = 3.0
        b[3::10] -= 2.5                                                      # empty (r < 0)
        A[4::10, m // 2:] = 0.0                                              # zero rows (b > 0: vacuous)
        if m >= 4 * d:                                                       # duplicated cube faces: degenerate vertices
            I = np.vstack([np.eye(d), -np.eye(d), np.eye(d), -np.eye(d)])
            A[5::10, :4 * d] = I[None]
            b[5::10, :4 * d] = np.r_[np.ones(d), np.zeros(d), np.ones(d), np.zeros(d)]
        ms = rng.integers(max(1, m - 5), m + 1, B).astype(np.int32)
        monkeypatch.setenv("PLP_CHEBY_WIDE", "0")
        ref = pa.cheby_ball_batch(A, b, m=ms)
        monkeypatch.setenv("PLP_CHEBY_WIDE", "1")
        got = pa.cheby_ball_batch(A, b, m=ms)
        monkeypatch.delenv("PLP_CHEBY_WIDE")
        assert np.array_equal(got["status"], ref["status"]), (m, d)
        ok = got["status"] == 0
        assert np.allclose(got["r"][ok], ref["r"][ok], rtol=0, atol=1e-12), (m, d)
        assert {0, 3} >= set(np.unique(got["status"])) or m < d + 1
        for k in range(0, B, 3):
            so, ro, _ = oracle.cheby(A[k, :ms[k]], b[k, :ms[k]])
            assert got["status"][k] == so, (m, d, k, got["status"][k], so)
            if so == 0:
                assert abs(got["r"][k] - ro) <= TOL * max(1.0, abs(ro)), (m, d, k)
                nrm = np.linalg.norm(A[k, :ms[k]], axis=1)
                assert np.max(A[k, :ms[k]] @ got["xc"][k] + nrm * got["r"][k] - b[k, :ms[k]]) <= 1e-8
    # the default dispatch takes it for the large shapes
    A = rng.standard_normal((50, 64, 16))
    A /= np.linalg.norm(A, axis=2, keepdims=True)
    b = 1.0 + rng.random((50, 64))
    dflt = pa.cheby_ball_batch(A, b)
    monkeypatch.setenv("PLP_CHEBY_WIDE", "1")
    forced = pa.cheby_ball_batch(A, b)
    assert np.array_equal(dflt["r"], forced["r"], equal_nan=True) and np.array_equal(dflt["status"], forced["status"])




def test_bench_kernel_full_size_repeatable_and_exact(pa, oracle):
    """BASELINE config 2 at full size (100 000 polytopes, m = 16, d = 3): eight launches in flight on two streams (every
    SIMD holding its four wavefronts, tails overlapping heads) return the bits of the first one, and a 3000-polytope
    sample of it equals the oracle in keep / flags / nlp.  (An instrumented build of round 3 that spilled 76 B per lane
    returned garbage for ~9 % of the tiles whenever more than one wavefront shared a SIMD; the shipped kernel is pinned
    here against that failure mode.)"""
    import torch
    from polytope_amd.synth import random_hpolytopes
    A, b = random_hpolytopes(100000, 16, 3, seed=2, stream=1)
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    ref = pa.reduce_batch(At, bt)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for rep in range(8):
        with torch.cuda.stream(streams[rep & 1]):
            outs.append(pa.reduce_batch(At, bt))
    torch.cuda.synchronize()
    for r in outs:
        for k in ("keep", "flags", "nlp"):
            assert torch.equal(r[k], ref[k]), k
        assert torch.equal(r["r"].view(torch.int64), ref["r"].view(torch.int64))
    keep = pa.keep_to_bool(ref["keep"].cpu().numpy(), 16)
    nlp, fl = ref["nlp"].cpu().numpy(), ref["flags"].cpu().numpy()
    for k in list(range(0, 1500)) + list(range(98500, 100000)):   # full tiles at the front, half-size tiles at the end
        o = oracle.reduce(A[k], b[k])
        assert np.array_equal(keep[k], o["keep"]) and int(nlp[k]) == o["nlp"] and int(fl[k]) == o["flags"], k


# ------------------------------------------------------------------------------ a slice of the soaks
@pytest.mark.parametrize("fam", ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"])
def test_soak_families_small(pa, oracle, fam):
    """A slice of scripts/soak_lane.py / soak_wide.py in the suite the driver runs: ~3 000 polytopes per family over nine shapes
    (d = 1..16, up to 64 rows: every kernel family of the dispatch), EVERY polytope against the oracle --
      fused reduce: keep mask, flags, LP count exact, radius 1e-9 (LP counts that differ on a prefilter tie are classified by
      scripts/soak_lane.py: prefilter_tie -- and must stay below one in a thousand);
      stand-alone Chebyshev ball: status exact, radius 1e-9;
      stand-alone bounding box: +-inf in the same places, finite sides within 1e-9 of the box's extent (1e-8 on UNBOUNDED polytopes
      of `dup`: scripts/soak_lane.py, box_equal, says why).
    `dup` is the family bounding_box went wrong on in round 5 (rows 1e-16 .. 1e-5 rad apart, no dedupe in front of its LPs)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_lane as SL
    rng = np.random.default_rng({"random": 11, "ragged": 12, "unbounded": 13, "dup": 14, "scaled": 15, "flat": 16, "lattice": 17}[fam])
    n = n_tie = 0
    for (d, m, B) in ((1, 6, 200), (2, 12, 500), (3, 16, 700), (3, 30, 400), (4, 22, 400), (6, 32, 300), (8, 64, 150), (12, 40, 150),
                      (16, 64, 100)):
        A, b, mr = SL.make(rng, B, m, d, fam)
        At, bt, mt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda(), torch.as_tensor(mr).cuda()
        rd = pa.reduce_batch(At, bt, mt)
        ch = pa.cheby_ball_batch(At, bt, m=mt)
        bb = pa.bbox_batch(At, bt, mt)
        keep = rd["keep"].cpu().numpy().view(np.uint64)
        flags, nlp, r = rd["flags"].cpu().numpy(), rd["nlp"].cpu().numpy(), rd["r"].cpu().numpy()
        cs, cr = ch["status"].cpu().numpy(), ch["r"].cpu().numpy()
        st, lb, ub = bb["status"].cpu().numpy(), bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy()
        for k in range(B):
            Ak, bk = A[k, :mr[k]], b[k, :mr[k]]
            o = oracle.reduce(Ak, bk)
            same = int(keep[k]) == int(o["mask"]) and int(flags[k]) == int(o["flags"]) and abs(r[k] - o["r"]) <= 1e-9 * max(1.0, abs(o["r"]))
            assert same, (fam, d, m, k, hex(int(keep[k])), hex(int(o["mask"])), int(flags[k]), o["flags"], r[k], o["r"])
            if int(nlp[k]) != int(o["nlp"]):
                assert SL.prefilter_tie(Ak, bk), (fam, d, m, k, int(nlp[k]), o["nlp"])
                n_tie += 1
            so, ro, _ = oracle.cheby(Ak, bk)
            assert int(cs[k]) == so and (so != 0 or abs(cr[k] - ro) <= 1e-9 * max(1.0, abs(ro))), (fam, d, m, k, int(cs[k]), so, cr[k], ro)
            if st[k] == 0:
                lo, hi, bad = oracle.bounding_box(Ak, bk)
                assert bad == 0 and SL.box_equal(lb[k], ub[k], lo, hi, hair_unbounded_tol=(1e-8 if fam == "dup" else None)), (fam, d, m, k, lb[k], lo, ub[k], hi)
            n += 1
    assert n_tie * 1000 <= n, (fam, n_tie, n)


def test_verifier_counters_and_histograms(pa):
    """plp_verify_counters / lp_histograms (counters): on random polytopes every answer certifies -- nothing
    goes to the careful engine --, on rows a hair apart and on unbounded LPs some do; the status / pivot histograms add up."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_lane as SL
    rng = np.random.default_rng(5)
    for fam, expect_zero in (("random", True), ("dup", False), ("unbounded", False)):
        A, b, mr = SL.make(rng, 2000, 20, 4, fam)
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        pa.cheby_ball_batch(At, bt)
        n_ch = pa.verify_careful_lps()
        pa.bbox_batch(At, bt)
        n_bb = pa.verify_careful_lps()
        c = torch.as_tensor(rng.standard_normal((2000, 4))).cuda()
        res = pa.lpsolve_batch(c, At, bt)
        n_lp = pa.verify_careful_lps()
        h = pa.lp_histograms(res)
        assert sum(h["status"].values()) == 2000 and sum(h["iters"][1]) == 2000
        if expect_zero:
            assert (n_ch, n_bb, n_lp) == (0, 0, 0), (fam, n_ch, n_bb, n_lp)
            assert h["status"] == {0: 2000}
        else:
            assert n_ch + n_bb + n_lp > 0, fam
        if fam == "unbounded":
            assert n_lp >= h["status"].get(3, 0) > 0      # every unbounded LP is the careful engine's verdict


def test_reduce_on_rows_a_hair_apart_reference_fixture(pa):
    """g23 (tests/golden/make_golden_reduce_dup.py): the REFERENCE's reduce() on 336 polytopes with rows a hair apart, elongated
    and shifted ones, d = 2..8 -- the engines' absolute tolerances against the real reference, not the oracle that shares them.
    Through the public reduce() on the 'hip' backend: the same rows kept (rows that coincide to 1e-12 stand for one another),
    the same emptiness verdict -- including the polytope whose Chebyshev LP the fused kernel ends "unbounded" (RF_F1OPEN: the
    host layer consults the verified ball and reduces it through the verified LPs) --, on every case the reference itself is
    consistent on (not: one polytope where HiGHS ends a redundancy LP with "numerical difficulties" and drops a facet)."""
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    from test_oracle_golden import _twin_classes
    g = load_golden("g23_reduce_dup.npz")
    old = solvers.default_solver
    solvers.default_solver = "hip"
    n = 0
    try:
        for i in range(len(g["m"])):
            if not g["pinned"][i]:
                continue
            m, d = int(g["m"][i]), int(g["d"][i])
            A, b = g["A"][i, :m * d].reshape(m, d).copy(), g["b"][i, :m].copy()
            q = pc.reduce(pc.Polytope(A, b, normalize=False))
            assert (q.A.size == 0) == bool(g["empty"][i]), (i, str(g["fam"][i]))
            if g["empty"][i]:
                continue
            # which input rows came back (rows are renormalised by the constructor)
            nrm = np.sqrt((A * A).sum(1))
            An, bn = A / nrm[:, None], b / nrm
            keep = np.zeros(m, bool)
            for a, bb in zip(q.A, q.b):
                dist = np.abs(An - a).max(1) + np.abs(bn - bb) / max(1.0, abs(bb))
                best = int(np.argmin(dist))      # (the closest input row: shifted copies lie 1e-7 apart, rounding 1e-13)
                assert dist[best] < 1e-9, (i, a, bb, dist[best])
                keep[best] = True
            assert np.array_equal(_twin_classes(A, b, keep), _twin_classes(A, b, g["keep"][i, :m])), (
                i, str(g["fam"][i]), d, m, np.nonzero(keep)[0], np.nonzero(g["keep"][i, :m])[0])
            n += 1
    finally:
        solvers.default_solver = old
    assert n >= 300
