#!/usr/bin/env python3
"""Secondary measurements at the BASELINE.json config sizes (bench.py stays the headline):

  C3  containment: 1M points x 10k polytopes (d=6, m=16), Region.contains semantics
  C5  quickhull assignment/furthest: 1M points, d=8, F in {9, 64, 512}
  LP  stand-alone batches: Chebyshev F1 (100k x (16,3)), generic lpsolve batch F2 (100k x (16,3)),
      (64,16) Chebyshev

One JSON line per measurement: time per launch from HIP events on the launch stream, roofline
(algorithmic bytes or flops of SURVEY.md 8(d) / peak) and a bounded CPU baseline
(numpy as the reference computes it, polytope/polytope.py:217-218, quickhull.py:117-121).
Usage (GPU box): python scripts/bench_configs.py [c3 c5 lp]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import polytope_amd as pa  # noqa: E402
from polytope_amd import solvers, synth  # noqa: E402

solvers.default_solver = "hip"  # the engine is opt-in (the default follows the reference's rule: scipy)

HBM_PEAK = 8000.0   # GB/s   (MI355X_MICROARCH.md)
FP64_PEAK = 78.6    # TFLOP/s vector = matrix on MI355X
dev = torch.device("cuda:0")


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def c3():
    P, N, d, m = 10000, 1000000, 6, 16
    A, b, X = synth.containment_workload(P, N, d=d, m=m, seed=0)
    At, bt, Xt = (torch.as_tensor(v).to(dev) for v in (A, b, X))
    out = pa.contains_batch(At, bt, Xt, 1e-7)
    ms = timeit(lambda: pa.contains_batch(At, bt, Xt, 1e-7), reps=5, warm=1)
    inside = int(out.sum().item())
    flops = 2.0 * m * d * N * P
    # CPU: numpy exactly as Polytope.contains on a 16-polytope sample, extrapolated
    t0 = time.perf_counter()
    acc = np.zeros(N, bool)
    for p in range(16):
        acc |= np.all(A[p].dot(X) - b[p][:, None] < 1e-7, axis=0)
    tcpu = (time.perf_counter() - t0) / 16 * P
    # spot parity at full size: first 64 polytopes on the first 200k points vs numpy
    ref = np.zeros(200000, bool)
    for p in range(P):
        if p < 64:
            ref |= np.all(A[p].dot(X[:, :200000]) - b[p][:, None] < 1e-7, axis=0)
    sub = pa.contains_batch(At[:64], bt[:64], Xt[:, :200000].contiguous(), 1e-7).cpu().numpy().astype(bool)
    print(json.dumps({"config": "C3 contains 1M points x 10k polytopes d=6 m=16", "ms": ms,
                      "tests_per_s": N * P / (ms * 1e-3), "points_inside": inside,
                      "roofline": {"bound": "fp64-valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP64_PEAK,
                                   "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / FP64_PEAK},
                      "hbm_algorithmic_GBs": (8 * d * N + 8 * P * m * (d + 1) + N) / (ms * 1e-3) / 1e9,
                      "cpu_baseline": {"numpy_1thread_s_extrapolated": tcpu, "speedup": tcpu / (ms * 1e-3)},
                      "spot_parity_equal": bool(np.array_equal(sub, ref))}), flush=True)


def c5():
    N, d = 1000000, 8
    for F in (9, 64, 512):
        X, nrm, off = synth.quickhull_workload(N, d=d, F=F, seed=0)
        Xt, nt, ot = (torch.as_tensor(v).to(dev) for v in (X, nrm, off))
        res = pa.assign_batch(Xt, nt, ot, 1e-7)
        ms = timeit(lambda: pa.assign_batch(Xt, nt, ot, 1e-7))
        bytes_alg = 8 * d * N + 12 * N + 8 * F * (d + 1)
        t0 = time.perf_counter()
        D = X @ nrm.T - off  # numpy baseline: all distances, first facet over tol, per-facet argmax
        hit = D > 1e-7
        fop = np.where(hit.any(1), hit.argmax(1), -1)
        tcpu = time.perf_counter() - t0
        same = bool(np.array_equal(res["facet"].cpu().numpy(), fop))
        print(json.dumps({"config": "C5 assign/furthest 1M points d=8 F=%d" % F, "ms": ms,
                          "point_facet_evals_per_s": N * F / (ms * 1e-3),
                          "assigned": int((res["facet"] >= 0).sum().item()),
                          "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK,
                                       "unit": "GB/s", "frac": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK},
                          "cpu_baseline": {"numpy_matmul_s": tcpu, "speedup": tcpu / (ms * 1e-3)},
                          "facet_ids_equal_numpy": same}), flush=True)


def lp():
    for (B, m, d) in [(100000, 16, 3), (20000, 64, 16), (20000, 64, 12), (20000, 48, 9)]:
        A, b = synth.random_hpolytopes(B, m, d, seed=1)
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        ms = timeit(lambda: pa.cheby_ball_batch(At, bt))
        by = B * (8 * (m * (d + 1) + m + d + 1) + 8 * (d + 2) + 4)
        print(json.dumps({"config": "F1 cheby batch B=%d m=%d d=%d" % (B, m, d), "ms": ms, "lp_per_s": B / (ms * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK,
                                       "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK}}), flush=True)
        # generic lpsolve batch on the F2 form of row 0 (h[0] += 0.1)
        c = torch.as_tensor(-A[:, 0, :].copy()).to(dev)
        h = b.copy()
        h[:, 0] += 0.1
        ht = torch.as_tensor(h).to(dev)
        ms = timeit(lambda: pa.lpsolve_batch(c, At, ht))
        by = B * (8 * (m * d + m + d) + 8 * (d + 1) + 4)
        print(json.dumps({"config": "lpsolve batch (F2 form) B=%d m=%d n=%d" % (B, m, d), "ms": ms,
                          "lp_per_s": B / (ms * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK,
                                       "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK}}), flush=True)
        # generic lpsolve batch in ORIGINAL coordinates (the polytope moved off the origin: phase 1 needed, as for the
        # reference's own calls), random cost
        rng = np.random.default_rng(3)
        h2 = torch.as_tensor(b + np.einsum("bij,bj->bi", A, rng.standard_normal((B, d)) * 3.0)).to(dev)
        c2 = torch.as_tensor(rng.standard_normal((B, d))).to(dev)
        res = pa.lpsolve_batch(c2, At, h2)
        ms = timeit(lambda: pa.lpsolve_batch(c2, At, h2))
        print(json.dumps({"config": "lpsolve batch (two-phase) B=%d m=%d n=%d" % (B, m, d), "ms": ms,
                          "lp_per_s": B / (ms * 1e-3), "mean_pivots": float(res["iters"].float().mean()),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK,
                                       "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK}}), flush=True)


def red():
    """Fused reduce() over the supported envelope: shapes other than the headline (16,3), up to (64,16)."""
    for (B, m, d) in [(100000, 16, 3), (50000, 16, 4), (20000, 32, 6), (20000, 32, 8), (5000, 64, 8), (5000, 64, 12),
                      (5000, 64, 16)]:
        A, b = synth.random_hpolytopes(B, m, d, seed=2)
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        res = pa.reduce_batch(At, bt)
        nlp = int(res["nlp"].sum().item())
        ms = timeit(lambda: pa.reduce_batch(At, bt), reps=5, warm=1)
        by = B * (8 * m * (d + 1) + 12)
        print(json.dumps({"config": "fused reduce B=%d m=%d d=%d" % (B, m, d), "ms": ms, "lps": nlp,
                          "lp_per_s": nlp / (ms * 1e-3), "kept_rows_mean": float(
                              sum(bin(int(k) & (2 ** 64 - 1)).count("1") for k in res["keep"].cpu().numpy()[:2000]) / 2000.0),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK,
                                       "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK}}), flush=True)


def c2s():
    """The (16,3) fused reduce below the headline batch size: the latency forms of reduce_lane_kernel (8 / 4 polytopes per
    wavefront) -- the shard of a strong-scaling step, the chunks of the end-to-end path, single calls."""
    full = synth.random_hpolytopes(100000, 16, 3, seed=0)
    for B in (50000, 25000, 12500, 6000, 1000, 64, 1):
        At, bt = torch.as_tensor(full[0][:B]).to(dev), torch.as_tensor(full[1][:B]).to(dev)
        res = pa.reduce_batch(At, bt)
        nlp = int(res["nlp"].sum().item())
        ms = timeit(lambda: pa.reduce_batch(At, bt), reps=21, warm=3)
        by = B * (8 * 16 * 4 + 12)
        print(json.dumps({"config": "fused reduce B=%d m=16 d=3" % B, "ms": ms, "lps": nlp, "lp_per_s": nlp / (ms * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                                       "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK}}), flush=True)


def bbox():
    """bounding_box batches: the fused kernel (Chebyshev LP + 2d LPs from its centre) against the 2d generic LPs."""
    import numpy as np
    for (B, m, d) in [(100000, 16, 3), (20000, 32, 6), (5000, 64, 8)]:
        A, b = synth.random_hpolytopes(B, m, d, seed=3)
        b = b + np.einsum("bij,bj->bi", A, np.random.default_rng(1).standard_normal((B, d)))  # origin outside
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        ms = timeit(lambda: pa.bbox_batch(At, bt))
        cost = torch.as_tensor(np.tile(np.vstack([np.eye(d), -np.eye(d)]), (B, 1))).to(dev)
        A2, b2 = At.repeat_interleave(2 * d, dim=0), bt.repeat_interleave(2 * d, dim=0)
        ms_gen = timeit(lambda: pa.lpsolve_batch(cost, A2, b2), reps=5, warm=1)
        print(json.dumps({"config": "bounding boxes B=%d m=%d d=%d" % (B, m, d), "ms": ms,
                          "lp_per_s": B * (2 * d + 1) / (ms * 1e-3), "generic_2d_lps_ms": ms_gen,
                          "speedup": ms_gen / ms}), flush=True)


def c4():
    """Config 4: 1000-cell Region in d=4 (10x10x5x2 grid of boxes on [0,1]^4): adjacency of all
    499 500 cell pairs (one batch of (16,5) Chebyshev LPs) and region_diff of a polytope against
    the whole Region (host DFS, every scan one batch)."""
    import itertools
    import polytope_amd.polytope as pc
    from polytope_amd import prop2partition as p2p
    shape = (10, 10, 5, 2)
    cells, index = [], []
    for idx in itertools.product(*[range(n) for n in shape]):
        cells.append(pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]))
        index.append(idx)
    index = np.array(index)
    t0 = time.perf_counter()
    adj = p2p.adjacency_matrix_dense(cells)
    t_adj = time.perf_counter() - t0
    t0 = time.perf_counter()
    adj = p2p.adjacency_matrix_dense(cells)
    t_adj2 = time.perf_counter() - t0
    want = (np.abs(index[:, None, :] - index[None, :, :]).max(axis=2) <= 1).astype(np.int8)
    npairs = len(cells) * (len(cells) - 1) // 2
    print(json.dumps({"config": "C4 find_adjacent_regions 1000 cells d=4", "pairs": npairs, "s_first": t_adj,
                      "s": t_adj2, "pair_lps_per_s": npairs / t_adj2,
                      "equals_grid_neighbourhood": bool(np.array_equal(adj, want))}), flush=True)
    from polytope_amd import batch
    stats = {}
    orig = batch.region_diff_search

    def spy(*a, **k):
        t = time.perf_counter()
        out = orig(*a, **k)
        stats.update(out[1], search_s=time.perf_counter() - t)
        return out
    batch.region_diff_search = spy
    half = [c for c, idx in zip(cells, index) if idx[0] < 5]  # the 500 cells with x0 < 0.5
    A, b = synth.random_hpolytopes(1, 12, 4, seed=4, bounded=True)
    P = pc.Polytope(A[0], 0.1 * b[0] + A[0] @ (0.5 * np.ones(4)))
    pc.region_diff(P.copy(), pc.Region(half))
    t0 = time.perf_counter()
    D = pc.region_diff(P.copy(), pc.Region(half))
    t_diff = time.perf_counter() - t0
    r, _ = pc.cheby_ball(P)
    print(json.dumps({"config": "C4 region_diff P(m=12,d=4, r=0.13) minus the 500 cells with x0<0.5 of the 1000-cell grid",
                      "s": t_diff, "library_search": dict(stats),
                      "pieces": len(D) if isinstance(D, pc.Region) else int(D.A.size > 0), "P_radius": float(r)}),
          flush=True)
    # the fixture case (tests/golden g12): P(m=16) of radius ~0.3-0.6, the reference needs 99 039 LPs and 186 s
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    g = load_golden("g12_config4.npz")
    P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
    pc.region_diff(P.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
    t0 = time.perf_counter()
    D = pc.region_diff(P.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
    t_diff = time.perf_counter() - t0
    print(json.dumps({"config": "C4 region_diff, fixture g12 (234 pieces; reference: 99 039 LPs, 186 s)", "s": t_diff,
                      "library_search": dict(stats), "pieces": len(D), "reference_lps": int(g["c4_diff_nlp"]),
                      "lps_per_s_reference_count": int(g["c4_diff_nlp"]) / t_diff}), flush=True)


def hull():
    """End-to-end quickhull with device-resident outside sets (polytope_amd.quickhull) beside qhull's C
    library (scipy.spatial.ConvexHull) on the same points; per-iteration cost = one plp_hull_reassign."""
    from scipy.spatial import ConvexHull
    from polytope_amd.quickhull import quickhull
    for (N, d) in [(1000000, 3), (1000000, 2), (200000, 4), (100000, 5), (1000000, 8)]:
        rng = np.random.default_rng(N + d)
        P = rng.standard_normal((N, d))
        if d == 8:  # a full d=8 hull of 1M Gaussian points has ~1e6 facets: time single passes instead
            from polytope_amd.batch import HullSession
            h = HullSession(P)
            nrm = rng.standard_normal((9, d))
            nrm /= np.linalg.norm(nrm, axis=1)[:, None]
            t0 = time.perf_counter()
            id0, cnt, am, mx = h.reassign([0], nrm, np.full(9, 1.0))
            t1 = time.perf_counter()
            _, cnt2, _, _ = h.reassign([id0], nrm, np.full(9, 1.5))
            t2 = time.perf_counter()
            h.close()
            print(json.dumps({"config": "C5 hull pass N=%d d=%d F=9 (host call incl. sync + D2H of results)" % (N, d),
                              "first_pass_ms": (t1 - t0) * 1e3, "repool_pass_ms": (t2 - t1) * 1e3,
                              "assigned": int(cnt.sum()), "repooled": int(cnt2.sum())}))
            continue
        np.random.seed(0)
        quickhull(P[:1000])  # warm the library
        np.random.seed(0)
        t0 = time.perf_counter()
        A, b, V = quickhull(P)
        t1 = time.perf_counter()
        ch = ConvexHull(P)
        t2 = time.perf_counter()
        same = np.array_equal(np.sort(V, axis=0), np.sort(P[np.unique(ch.vertices)], axis=0))
        print(json.dumps({"config": "quickhull N=%d d=%d" % (N, d), "facets": int(A.shape[0]), "vertices": int(V.shape[0]),
                          "hip_s": t1 - t0, "us_per_facet": (t1 - t0) / A.shape[0] * 1e6,
                          "cpu_baseline": {"kind": "scipy.spatial.ConvexHull (qhull C library, 1 core)", "s": t2 - t1},
                          "same_vertex_set": bool(same)}))


def lat():
    """Small batches (what single-polytope calls of the Python layer issue): device time per call of the fused reduce
    and the fused bounding boxes, latency form (default) against the batch form, outputs compared bitwise."""
    import os
    for (m, d) in ((16, 3), (32, 6), (64, 8), (32, 12)):
        for B in (1, 256, 4096):
            A, b = synth.random_hpolytopes(B, m, d, seed=3, stream=0)
            At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
            rec = {"config": "small batch B=%d m=%d d=%d" % (B, m, d)}
            for name, fn, env in (("reduce", lambda: pa.reduce_batch(At, bt), "PLP_REDUCE_SPLIT"),
                                  ("bbox", lambda: pa.bbox_batch(At, bt), "PLP_BBOX_SPLIT")):
                outs = {}
                for form, val in (("batch_form", "0"), ("latency_form", "1")):
                    os.environ[env] = val
                    outs[form] = fn()
                    rec["%s_%s_us" % (name, form)] = timeit(fn, reps=50, warm=5) * 1e3
                os.environ.pop(env)
                rec["%s_bitwise_equal" % name] = bool(all(torch.equal(outs["batch_form"][k].view(torch.uint8),
                                                                      outs["latency_form"][k].view(torch.uint8))
                                                          for k in outs["batch_form"]))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c5", "lp", "red", "bbox"]
    for w in which:
        globals()[w]()
