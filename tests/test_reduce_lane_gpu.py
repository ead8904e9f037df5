"""GPU: reduce_lane_kernel (polytope_amd/csrc/plp_reduce_lane.hip) -- the fused reduce() (polytope/polytope.py:1053-1163) of
polytopes with up to 32 rows in d <= 3 with the box LPs (:1118-1134) and the redundancy LPs (:1142-1160) solved one LP per
lane (plp_lane_lp.hpp).  Parity bar: keep mask, flags and LP count equal to the oracle's and to the lane-group kernels',
r / xc bit for bit the lane-group kernels' (the same F1), every tile shape bit for bit the same, and the polytopes the
fast path hands back redone inside the kernel with the same result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu

_SWITCHES = ("PLP_REDUCE_LANE", "PLP_REDUCE_LANE_GS", "PLP_REDUCE_LANE_MIX", "PLP_REDUCE_RETRY_ALL")


@pytest.fixture(scope="module")
def pa():
    import polytope_amd as pa
    from polytope_amd import _lib
    assert _lib.available(), "libplp_hip.so did not load or no gfx950 device: the HIP path is mandatory"
    return pa


@pytest.fixture
def run(pa):
    import torch

    def go(A, b, m=None, **env):
        for k in _SWITCHES:
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            t = lambda v, dt=None: None if v is None else torch.as_tensor(np.ascontiguousarray(v, dtype=dt)).cuda()  # noqa: E731
            res = pa.reduce_batch(t(A), t(b), t(m, np.int32))
            torch.cuda.synchronize()
            return {k: v.cpu().numpy() for k, v in res.items()}
        finally:
            for k in _SWITCHES:
                os.environ.pop(k, None)
    return go


def _same_bits(x, y, keys=("keep", "flags", "nlp", "r", "xc")):
    return all(np.array_equal(x[k].view(np.uint8), y[k].view(np.uint8)) for k in keys)


def _same(x, y):
    """Verdicts exact; the Chebyshev ball to rounding.  (F1 is the lane-group simplex in every tile shape, but where its ratio
    test TIES -- duplicated rows, cubes -- the pivot it takes depends on how the rows lie on the lanes: 4 / 2 / 1 per lane.
    The ball is the same, its last bits are not; on data without ties -- the random batches -- every bit is equal, which the
    tests below check with _same_bits.)"""
    return _same_bits(x, y, keys=("keep", "flags", "nlp")) and \
        np.allclose(x["r"], y["r"], rtol=0.0, atol=1e-12, equal_nan=True) and \
        np.allclose(x["xc"], y["xc"], rtol=0.0, atol=1e-9, equal_nan=True)


def _vs_oracle(oracle, res, A, b, m=None):
    if m is None:
        R = oracle.reduce_batch(A, b)
        return (np.array_equal(res["keep"].view(np.uint64), R["keep"]) and np.array_equal(res["flags"], R["flags"])
                and np.array_equal(res["nlp"], R["nlp"]) and bool(np.all(np.abs(res["r"] - R["r"]) <= 1e-9)))
    for k in range(A.shape[0]):
        o = oracle.reduce(A[k, :m[k]], b[k, :m[k]])
        if o["mask"] != int(res["keep"][k]) or o["flags"] != int(res["flags"][k]) or o["nlp"] != int(res["nlp"][k]) \
                or not abs(o["r"] - res["r"][k]) <= 1e-9:
            return False
    return True


def _spoil(A, b, rng):
    """duplicated / shifted-parallel rows and an infeasible row here and there"""
    B, m, _ = A.shape
    for k in range(0, B, 5):
        j = int(rng.integers(m))
        A[k, (j + 1) % m] = A[k, j]
        b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
    for k in range(3, B, 11):
        b[k, 0] = -4.0


def test_lane_kernel_equals_oracle_and_lane_group_kernels(run, oracle):
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(17)
    for (m, d) in [(16, 3), (12, 3), (7, 3), (3, 3), (16, 2), (9, 2), (4, 2), (6, 1), (2, 1),
                   (32, 3), (24, 3), (17, 3), (32, 2), (21, 2), (20, 1)]:   # (17..32 rows: 32 row slots per polytope)
        for B in (1, 7, 130, 3000):
            A, b = random_hpolytopes(B, m, d, seed=13 * m + d + B, bounded=(B != 130))
            lane = run(A, b)
            assert _vs_oracle(oracle, lane, A, b), (m, d, B)
            assert _same_bits(lane, run(A, b, PLP_REDUCE_LANE=0)), (m, d, B)   # random rows: no ties, every bit
            _spoil(A, b, rng)
            lane = run(A, b)
            assert _vs_oracle(oracle, lane, A, b), (m, d, B, "spoiled")
            assert _same(lane, run(A, b, PLP_REDUCE_LANE=0)), (m, d, B, "spoiled")
            rows = rng.integers(max(1, m - 5), m + 1, B).astype(np.int32)
            lane = run(A, b, rows)
            assert _vs_oracle(oracle, lane, A, b, rows), (m, d, B, "ragged")
            assert _same(lane, run(A, b, rows, PLP_REDUCE_LANE=0)), (m, d, B, "ragged")


def test_lane_tile_shapes_bit_for_bit(run, oracle):
    """16 / 8 / 4 polytopes per wavefront (the LPs of a polytope on single lanes, pairs, quads), the mixed launch and the
    shape the dispatch picks, at batch sizes on both sides of every switch-over: every output bit the same on random rows;
    with duplicated / infeasible rows mixed in the verdicts are the same and the ball agrees to rounding (see _same)."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(4)
    for (B, m, d) in [(1, 16, 3), (77, 16, 3), (9000, 16, 3), (14003, 13, 3), (41000, 16, 3), (70001, 16, 3), (5000, 16, 2),
                      (3000, 5, 1), (9, 32, 3), (9000, 32, 3), (17000, 24, 2), (30001, 19, 3)]:
        A, b = random_hpolytopes(B, m, d, seed=B + m)
        for spoiled in (False, True):
            if spoiled:
                _spoil(A, b, rng)
            ref = run(A, b, PLP_REDUCE_LANE_GS=4 if m <= 16 else 8)   # (17..32 rows: 8 or 4 polytopes per wavefront)
            if B <= 20000:
                assert _vs_oracle(oracle, ref, A, b), (B, m, d)
            for env in ({"PLP_REDUCE_LANE_GS": 8}, {"PLP_REDUCE_LANE_GS": 16}, {}, {"PLP_REDUCE_LANE_MIX": 0}, {"PLP_REDUCE_LANE_MIX": 24}):
                got = run(A, b, **env)
                assert (_same if spoiled else _same_bits)(got, ref), (B, m, d, env, spoiled)


def test_lane_handover_inside_the_kernel(run, oracle):
    """What the fast path hands back is redone by the general engine in the same tile (one launch): forced for every
    polytope (PLP_REDUCE_RETRY_ALL=1), and arising by itself on degenerate vertices (pyramids: Bland's rule territory)
    and on structured polytopes (ties, twins, tangent rows)."""
    from polytope_amd.synth import random_hpolytopes
    from structured_cases import structured_polytopes
    from test_gpu_parity import _pyramids
    rng = np.random.default_rng(23)
    for gs in (4, 8, 16):
        for (B, m, d) in [(3000, 16, 3), (700, 11, 2), (50, 4, 1), (900, 32, 3), (500, 23, 2)]:
            A, b = random_hpolytopes(B, m, d, seed=gs + B)
            _spoil(A, b, rng)
            forced = run(A, b, PLP_REDUCE_LANE_GS=gs, PLP_REDUCE_RETRY_ALL=1)
            assert _vs_oracle(oracle, forced, A, b), (gs, B, m, d)
            plain = run(A, b, PLP_REDUCE_LANE_GS=gs)
            assert _same_bits(forced, plain, keys=("keep", "flags", "nlp")), (gs, B, m, d)
    A, b = _pyramids(400, 16, 3, rng)
    for gs in (4, 8, 16):
        assert _vs_oracle(oracle, run(A, b, PLP_REDUCE_LANE_GS=gs), A, b), gs
    As, bs, _ = structured_polytopes(4096)
    ref = run(As, bs, PLP_REDUCE_LANE=0)
    for gs in (4, 8, 16):
        got = run(As, bs, PLP_REDUCE_LANE_GS=gs)
        assert _same_bits(got, ref, keys=("keep", "flags", "nlp")), gs
        assert _vs_oracle(oracle, got, As, bs), gs


def test_lane_kernel_at_d4(run, oracle):
    """d = 4 (walk4, plp_lane_lp.hpp): the dispatch takes it beyond 3 000 polytopes of 14..32 rows and beyond 40 000 of fewer; here it is forced
    (PLP_REDUCE_LANE=1) on small ones too -- verdicts equal to the oracle's and the lane-group kernels', every tile shape."""
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(41)
    for (B, m) in [(1, 16), (300, 16), (2000, 12), (1500, 9), (1200, 32), (700, 21), (33000, 16)]:
        A, b = random_hpolytopes(B, m, 4, seed=B + m, bounded=(B != 1500))
        _spoil(A, b, rng)
        groups = run(A, b, PLP_REDUCE_LANE=0)
        if B <= 2000:
            assert _vs_oracle(oracle, groups, A, b), (B, m)
        for env in ({"PLP_REDUCE_LANE": 1}, {"PLP_REDUCE_LANE": 1, "PLP_REDUCE_LANE_GS": 8}, {"PLP_REDUCE_LANE": 1, "PLP_REDUCE_LANE_GS": 16},
                    {"PLP_REDUCE_LANE": 1, "PLP_REDUCE_RETRY_ALL": 1}, {}):
            if B > 5000 and "PLP_REDUCE_RETRY_ALL" in env:
                continue
            assert _same(run(A, b, **env), groups), (B, m, env)


def test_bbox_lane_kernel(pa, oracle):
    """plp_bbox_batch at (<= 32 rows, d <= 3): the 2 d box LPs one LP per lane (bbox_lane_kernel) against the lane-group
    kernels (PLP_BBOX_LANE=0) and the oracle's generic LPs: the same boxes to 1e-9, +-inf in the same places, and the same
    status except for the rare polytope whose walk is handed back (status 1: the caller's generic LPs take it)."""
    import torch
    from polytope_amd.synth import random_hpolytopes
    rng = np.random.default_rng(6)
    for (B, m, d, bounded) in [(5000, 16, 3, True), (3000, 32, 3, True), (70000, 16, 3, True), (4000, 12, 2, True), (3000, 24, 2, False),
                               (2000, 7, 1, True), (1500, 9, 3, False), (1, 16, 3, True)]:
        A, b = random_hpolytopes(B, m, d, seed=m + d + B, bounded=bounded)
        b = b + np.einsum("bij,bj->bi", A, rng.standard_normal((B, d)))   # boxes away from the origin
        for k in range(3, B, 17):
            b[k, 0] = -40.0                                                # empty polytopes: status 1
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        out = {}
        for lane in ("0", "1"):
            os.environ["PLP_BBOX_LANE"] = lane
            try:
                out[lane] = {k: v.cpu().numpy() for k, v in pa.bbox_batch(At, bt).items()}
            finally:
                os.environ.pop("PLP_BBOX_LANE", None)
        g, ln = out["0"], out["1"]
        both = (g["status"] == 0) & (ln["status"] == 0)
        assert np.count_nonzero(g["status"] != ln["status"]) <= max(1, B // 5000), (B, m, d)
        assert not np.any((ln["status"] == 0) & (g["status"] == 1)), (B, m, d)   # never settles what the lane groups leave
        for key in ("lb", "ub"):
            x, y = g[key][both], ln[key][both]
            assert np.array_equal(np.isinf(x), np.isinf(y)) and np.array_equal(np.sign(x[np.isinf(x)]), np.sign(y[np.isinf(y)]))
            fin = np.isfinite(x)
            assert np.all(np.abs(x[fin] - y[fin]) <= 1e-9 * np.maximum(1.0, np.abs(x[fin]))), (B, m, d, key)
        for k in range(0, min(B, 400), 7):
            if ln["status"][k] == 0:
                lb, ub, bad = oracle.bounding_box(A[k], b[k])
                assert bad == 0
                for got, want in ((ln["lb"][k], lb.ravel()), (ln["ub"][k], ub.ravel())):
                    fin = np.isfinite(want)
                    assert np.array_equal(np.isfinite(got), fin) and np.all(np.abs(got[fin] - want[fin]) <= 1e-9 * np.maximum(1.0, np.abs(want[fin])))


def test_lane_kernel_reference_fixtures_in_batches(run):
    """The reference's own reduce() results (fixtures g2: up to 16 rows, g20: 17..32 rows incl. the stacks of
    Polytope.intersect; tests/golden/make_golden.py) through the lane kernel as BATCHES per shape, every tile shape:
    the reference's kept-row sets exactly, its Chebyshev radius to 1e-9."""
    from conftest import load_golden
    from polytope_amd import _lib
    for fixture in ("g2_reduce.npz", "g20_reduce_rows32.npz"):
        g = load_golden(fixture)
        shapes = sorted({(int(m), int(d)) for m, d in zip(g["m"], g["d"]) if d <= 3 and m <= 32})
        assert shapes
        for (m, d) in shapes:
            idx = [i for i in range(len(g["m"])) if int(g["m"][i]) == m and int(g["d"][i]) == d]
            A = np.stack([g["A"][i, :m * d].reshape(m, d) for i in idx])
            b = np.stack([g["b"][i, :m] for i in idx])
            for gs in (4, 8, 16):
                res = run(A, b, PLP_REDUCE_LANE=1, PLP_REDUCE_LANE_GS=gs)
                for k, i in enumerate(idx):
                    empty = bool(int(res["flags"][k]) & _lib.RF_EMPTY)
                    assert empty == bool(g["empty"][i]), (fixture, m, d, gs, i)
                    if empty:
                        continue
                    mask = np.array([(int(res["keep"][k]) >> r) & 1 for r in range(m)], dtype=bool)
                    assert np.array_equal(mask, g["mask"][i, :m]), (fixture, m, d, gs, i)
                    assert abs(res["r"][k] - g["r"][i]) <= 1e-9, (fixture, m, d, gs, i)


def test_lane_kernel_d4_rows_tilted_from_the_cost(run, oracle):
    """tests/golden/lane_w4_tilted.npz (see tests/test_lane_lp_host.py: nine (23,4) polytopes on which walk4 left its planes
    by 1.4e-4 and the fused reduce lost a facet to the prefilter): every tile shape of the lane kernel at d = 4 against the
    oracle and the lane-group kernels, the nine alone and scattered through a batch large enough for the default dispatch."""
    from conftest import load_golden
    from polytope_amd.synth import random_hpolytopes
    g = load_golden("lane_w4_tilted.npz")
    A9, b9, m9 = g["A"], g["b"], g["m"].astype(np.int32)
    for env in ({"PLP_REDUCE_LANE": 1}, {"PLP_REDUCE_LANE": 1, "PLP_REDUCE_LANE_GS": 8}, {"PLP_REDUCE_LANE": 1, "PLP_REDUCE_LANE_GS": 16},
                {"PLP_REDUCE_LANE": 0}):
        assert _vs_oracle(oracle, run(A9, b9, m9, **env), A9, b9, m9), env
    B = 31000
    A, b = random_hpolytopes(B, A9.shape[1], 4, seed=5)
    m = np.full(B, A9.shape[1], np.int32)
    at = np.arange(9) * 3301 + 17
    A[at], b[at], m[at] = A9, b9, m9
    lane = run(A, b, m)                       # beyond 30 000 polytopes: the lane kernel by default
    assert _same(lane, run(A, b, m, PLP_REDUCE_LANE=0))
    sub = {k: v[at] for k, v in lane.items()}
    assert _vs_oracle(oracle, sub, A9, b9, m9)
