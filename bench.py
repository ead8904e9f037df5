#!/usr/bin/env python3
"""bench.py -- LP solves/sec of the fused reduce() hot path on N MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
    batch of 100 000 random H-polytopes, d=3, m=16 facets; one step = one pass of
    reduce() over the whole batch = for every polytope 1 Chebyshev LP (F1) + 2d bounding-box
    LPs (F3) + one redundancy LP (F2) per row that survives dedupe/prefilter
    (reference: polytope/polytope.py:1053-1163).  value = LPs solved per second; the LP
    count is the number of lpsolve() calls the reference would issue on the same input
    (returned by the kernel as nlp[] and checked against the oracle in tests/).

The timed region rotates over NB = 6 distinct 100k-polytope batches (6 x 52.4 MB = 314 MB, more than the 256 MiB
Infinity Cache), all resident in HBM before the clock starts, so a step never re-reads what the previous step left
in cache.  Extra objects of the line (N = 1): `end_to_end` = the same pass for a caller that holds numpy arrays
(host buffers in, host-visible results out; pageable and pinned, H2D / kernel / D2H split) and `cpu_baseline` =
scipy.optimize.linprog called as polytope/solvers.py:152-154 calls it, on the host cores (count stated).

Launch:  python bench.py [--gpus N --steps K --warmup W].  For N > 1 either under a launcher,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
or plainly as `python bench.py --gpus N`: without WORLD_SIZE in the environment it starts its N ranks itself.
One rank per GPU; `value` = weak scaling (every rank reduces its own 100k-polytope batches, the packed results
-- 24 B per polytope -- are all-gathered over RCCL, coalesced and overlapped); the same run also measures ONE
100k-polytope batch partitioned over the ranks (north_star) and reports it as the object `strong`.

Timing: a region = the K steps repeated until it lasts >= 50 ms, between barrier + synchronize on both sides, max over
ranks; five regions, the median is reported (`--steps 20` alone would be a 4 ms window).  After the clocks every
polytope of every batch is reduced again by the CPU oracle on the host cores and compared (`parity_checked`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_PER_GPU, M_ROWS, DIM = 100000, 16, 3
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMD_CLOCK_HZ = 2.4e9  # MI355X_MICROARCH.md: peak engine clock


def cpu_baseline(A, b, nlp_gpu):
    """Rank 0, N=1 only, bounded samples of batch 0.  `value` = the reference's own LP path on this host:
    scipy.optimize.linprog called exactly as polytope/solvers.py:152-154 calls it (the only backend the reference
    finds installed here), on the redundancy LPs (F2) of a sample, one process per host core; the one-process
    figure and the C restatement (oracle/plp_oracle.c, 'port') on one core / all cores ride along as side keys.
    The oracle leg doubles as a check: the LP count the GPU reported for the sample must equal the oracle's."""
    from oracle import oracle as O
    O.build()
    out = {"value": None, "unit": "LP/s", "cores": None, "kind": "reference", "sample": ""}
    ncpu = os.cpu_count() or 1
    try:
        from scipy.optimize import linprog
        import multiprocessing as mp
        ns = 128  # x 16 rows = 2048 LPs on one process (a >= 2000-LP sample)
        t0 = time.perf_counter()
        cnt = 0
        for k in range(ns):
            Ak, bk = A[k], b[k].copy()
            for row in range(M_ROWS):
                h = bk.copy()
                h[row] += 0.1
                linprog(-Ak[row], Ak, h, None, None, bounds=(None, None))
                cnt += 1
        t1 = time.perf_counter() - t0
        out["scipy_linprog_1proc_lp_per_s"] = cnt / t1
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        # the all-cores sample: a pilot of two tasks per worker gives the rate, the timed sample is sized for >= 10 s of clock
        # (tasks of 4 polytopes = 64 LPs, ~40 ms: the pool's dispatch is noise)
        per_task = 4
        avail = (A.shape[0] - ns) // per_task
        mk = lambda lo, hi: [[(A[k], b[k]) for k in range(ns + t * per_task, ns + (t + 1) * per_task)] for t in range(lo, hi)]
        with mp.get_context("fork").Pool(ncpu) as pool:
            pool.map(_scipy_chunk, mk(0, min(ncpu, avail)), chunksize=1)  # start the workers, import scipy: not timed
            t0 = time.perf_counter()
            pilot = pool.map(_scipy_chunk, mk(0, min(2 * ncpu, avail)), chunksize=1)
            rate = sum(pilot) / (time.perf_counter() - t0)
            ntask = int(min(avail, max(2 * ncpu, 16.0 * rate / (M_ROWS * per_task))))   # (the pilot runs ~1.3 x the sample's rate: 16 s of it is >= 10 s of clock)
            chunks = mk(0, ntask)
            t0 = time.perf_counter()
            res = pool.map(_scipy_chunk, chunks, chunksize=1)
            t2 = time.perf_counter() - t0
        out["value"] = sum(res) / t2
        out["cores"] = ncpu
        out["sample"] = ("scipy.optimize.linprog (HiGHS), called as polytope/solvers.py:152-154, on the F2 LPs of %d "
                         "polytopes of batch 0 over %d processes (%d LPs, %.1f s, workers started and scipy imported before "
                         "the clock, BLAS pools held to one thread); 1 process: %d LPs, %.1f s"
                         % (len(chunks) * per_task, ncpu, sum(res), t2, cnt, t1))
    except Exception as e:  # report, never fail the bench on the baseline
        out["scipy_error"] = repr(e)
    n = 40000
    t0 = time.perf_counter()
    lps = 0
    for k in range(n):
        lps += O.reduce(A[k], b[k])["nlp"]
    t_or = time.perf_counter() - t0
    gpu_lps = int(nlp_gpu[:n].sum())
    assert gpu_lps == lps, "LP count of the GPU (%d) != oracle (%d) on the first %d polytopes" % (gpu_lps, lps, n)
    out["port_1core_lp_per_s"] = lps / t_or
    out["port_presolve"] = False
    out["port_note"] = ("the C port solves EVERY LP the reference issues with its simplex (no presolve): its LP/s compare with "
                        "`value` as work disposed of, and with config.value_simplex_only as simplex runs")
    out["nlp_checked"] = "GPU nlp[:%d].sum() == oracle count == %d" % (n, lps)
    out["sample"] += "; port: oracle/plp_oracle.c reduce() on the first %d polytopes (%d LPs, %.1f s)" % (n, lps, t_or)
    try:  # the same C port on every host core: what a competent CPU code does with the box
        import multiprocessing as mp
        global _ORACLE_AB
        _ORACLE_AB = (A, b)
        per = 2000
        tasks = [((t * per) % A.shape[0], per) for t in range(4 * ncpu)]
        with mp.get_context("fork").Pool(ncpu) as pool:
            pool.map(_oracle_chunk, tasks[:ncpu], chunksize=1)  # start the workers, load the library
            t0 = time.perf_counter()
            res = pool.map(_oracle_chunk, tasks, chunksize=1)
            t_all = time.perf_counter() - t0
        out["port_allcores_lp_per_s"] = sum(res) / t_all
        out["port_allcores_cores"] = ncpu
    except Exception as e:
        out["port_allcores_error"] = repr(e)
    try:
        with open("/proc/cpuinfo") as f:
            models = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")]
        out["cpu_model"] = models[0] if models else "?"
        out["os_cpu_count"] = os.cpu_count()
    except Exception:
        pass
    return out


def end_to_end(torch, pa, A, b, dev, reps=10):
    """The same pass for a caller whose polytopes live in host memory (wall time = host-visible results).
    (i) pageable numpy arrays through the C ABI's host entry point (copies in, kernel, copies out, blocks);
    (ii) pinned host tensors: chunked async H2D on a copy stream, the kernel of a chunk as soon as it has landed, one D2H
    of keep / flags / r / nlp, one synchronisation.  PCIe-inclusive figures (fraction of 64 GB/s stated), never `value`."""
    out = {"unit": "LP/s", "note": "numpy in -> host-visible keep/flags/r/xc/nlp out; never the headline value"}
    res = pa.reduce_batch(A, b)
    nlp = int(res["nlp"].sum())
    t0 = time.perf_counter()
    for _ in range(reps):
        res = pa.reduce_batch(A, b)
    t = (time.perf_counter() - t0) / reps
    out["pageable"] = {"ms_per_pass": t * 1e3, "value": nlp / t, "h2d_bytes": A.nbytes + b.nbytes,
                       "d2h_bytes": int(sum(v.nbytes for v in res.values()))}
    from polytope_amd.dist import ResultBuffer
    Ap, bp = torch.as_tensor(A).pin_memory(), torch.as_tensor(b).pin_memory()
    Ad, bd = torch.empty_like(Ap, device=dev), torch.empty_like(bp, device=dev)
    B = A.shape[0]
    # the kernel writes keep / r / flags / nlp straight into one flat 24 B-per-polytope buffer (the exchange buffer of
    # the multi-GPU path): ONE D2H copy brings them to the host
    rb = ResultBuffer(torch, B, A.shape[2], dev)
    host = torch.empty((rb.nbytes,), dtype=torch.uint8).pin_memory()
    WU = 3  # untimed passes: freshly pinned pages are touched for the first time
    # Chunked and overlapped, as the library does for pageable input (csrc/plp_stage.hpp): the batch crosses PCIe in NCH
    # chunks on a copy stream, the fused kernel of chunk c starts on the compute stream as soon as chunk c has landed
    # (an event per chunk) and writes into its slice of the result buffer; one D2H copy of the 24 B-per-polytope buffer
    # at the end.  The upload is the critical path; what remains of the kernels is the last chunk's.
    NCH = int(os.environ.get("PLP_BENCH_E2E_CHUNKS", "4"))   # (8 chunks with A and b copied per chunk: 16 copies, 44 GB/s instead of 53, no gain over one copy)
    step = ((B + NCH - 1) // NCH + 15) // 16 * 16   # whole tiles of 16 polytopes
    bounds = [(lo, min(B, lo + step)) for lo in range(0, B, step)]
    copy_st = torch.cuda.Stream(device=dev)
    main_st = torch.cuda.current_stream()
    landed = [torch.cuda.Event() for _ in bounds]
    views = [{k: v[lo:hi] for k, v in rb.views.items()} for lo, hi in bounds]
    ev_a = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(reps + WU)]   # upload, on the copy stream
    ev_k = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps + WU)]   # first kernel .. last kernel .. D2H
    walls = []
    for i in range(reps + WU):
        t0 = time.perf_counter()
        copy_st.wait_stream(main_st)
        with torch.cuda.stream(copy_st):
            ev_a[i][0].record()
            bd.copy_(bp, non_blocking=True)           # b (a quarter of the bytes) in one copy, then A chunk by chunk
            for c, (lo, hi) in enumerate(bounds):
                Ad[lo:hi].copy_(Ap[lo:hi], non_blocking=True)
                landed[c].record()
            ev_a[i][1].record()
        for c, (lo, hi) in enumerate(bounds):
            main_st.wait_event(landed[c])
            if c == 0:
                ev_k[i][0].record()
            pa.reduce_batch(Ad[lo:hi], bd[lo:hi], out=views[c])
        ev_k[i][1].record()
        host.copy_(rb.flat, non_blocking=True)
        ev_k[i][2].record()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    assert int(rb.split(host)[0]["nlp"].sum().item()) == nlp
    w = sorted(walls[WU:])[reps // 2]  # median
    med = lambda evl, a, c: sorted(evl[i][a].elapsed_time(evl[i][c]) for i in range(WU, reps + WU))[reps // 2]  # noqa: E731
    h2d_ms = med(ev_a, 0, 1)
    h2d_bytes = A.nbytes + b.nbytes
    out["pinned"] = {"ms_per_pass": w * 1e3, "value": nlp / w, "chunks": len(bounds), "h2d_ms": h2d_ms,
                     "kernels_first_to_last_ms": med(ev_k, 0, 1), "d2h_ms": med(ev_k, 1, 2),
                     "h2d_GBs": h2d_bytes / h2d_ms / 1e6, "d2h_bytes": int(rb.nbytes),
                     "pcie_frac_of_64GBs": (h2d_bytes + rb.nbytes) / w / 64e9,
                     "note": "upload in %d chunks on a copy stream, the kernel of a chunk starts when it has landed" % len(bounds)}
    out["pageable"]["pcie_frac_of_64GBs"] = (out["pageable"]["h2d_bytes"] + out["pageable"]["d2h_bytes"]) / t / 64e9
    return out


_ORACLE_AB = None



def secondary(torch, pa, dev):
    """The other BASELINE configs, driver-timed beside the headline (rank 0, N = 1, about three seconds in all; never `value`).
    Each with its own roofline figure (algorithmic flops / bytes) and a parity sample against the oracle:
      C3  1 M points x 10 k polytopes (d = 6, m = 16), Region.contains semantics (polytope/polytope.py:206-218, :732-746)
      C5  quickhull's distance / first-facet assignment / furthest point, 1 M points, d = 8, F = 9 / 64 / 512
          (polytope/quickhull.py:117-121, :224-245, :87-102)
      C4  the 1000-cell grid in d = 4: all 499 500 pair LPs of find_adjacent_regions (polytope/prop2partition.py:46-63), and
          region_diff of the fixture polytope against 500 cells (polytope/polytope.py:2117-2282; tests/golden/g12: 234 pieces)
      fused reduce at the far end of the envelope, (64,16) x 5 000 (polytope/polytope.py:1053-1163)"""
    import itertools
    import numpy as np
    from oracle import oracle as O
    from polytope_amd import synth
    out = {}

    def timeit(fn, reps, warm=1):
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

    # ---- C3
    P, N, d, m = 10000, 1000000, 6, 16
    A, b, X = synth.containment_workload(P, N, d=d, m=m, seed=0)
    At, bt, Xt = (torch.as_tensor(v).to(dev) for v in (A, b, X))
    got = pa.contains_batch(At, bt, Xt, 1e-7)
    ms = timeit(lambda: pa.contains_batch(At, bt, Xt, 1e-7), reps=3)
    flops = 2.0 * m * d * N * P
    rng = np.random.default_rng(0)
    pts = rng.choice(N, 1500, replace=False)     # 1 500 points against ALL polytopes on the oracle
    want = O.contains(A, b, np.ascontiguousarray(X[:, pts].T), region=True)
    out["C3_contains"] = {"workload": "1M points x 10k polytopes, d=6, m=16 (BASELINE configs[2])", "ms": ms,
                          "tests_per_s": N * P / (ms * 1e-3), "points_inside": int(got.sum().item()),
                          "roofline": {"bound": "fp64-valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                       "frac": flops / (ms * 1e-3) / 1e12 / 78.6},
                          "parity": {"oracle_points": 1500, "equal": bool(np.array_equal(got.cpu().numpy()[pts].astype(bool), want.astype(bool)))}}
    del At, bt, Xt, got
    # ---- C5
    c5 = {}
    for F in (9, 64, 512):
        X, nrm, off = synth.quickhull_workload(N, d=8, F=F, seed=0)
        Xt, nt, ot = (torch.as_tensor(v).to(dev) for v in (X, nrm, off))
        res = pa.assign_batch(Xt, nt, ot, 1e-7)
        ms = timeit(lambda: pa.assign_batch(Xt, nt, ot, 1e-7), reps=5)
        by = 8 * 8 * N + 12 * N + 8 * F * 9
        k = 20000                                  # the first 20 000 points on the oracle: facet ids, distances bit for bit
        fop, dist, _am, _mx = O.assign(X[:k], nrm, off, 1e-7)
        same = bool(np.array_equal(res["facet"][:k].cpu().numpy(), fop) and
                    np.array_equal(res["dist"][:k].cpu().numpy().view(np.int64), dist.view(np.int64)))
        c5["F=%d" % F] = {"ms": ms, "point_facet_evals_per_s": N * F / (ms * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                          "parity": {"oracle_points": k, "bitwise_equal": same}}
        del Xt
    out["C5_assign"] = dict(workload="1M points, d=8 (BASELINE configs[4])", **c5)
    # ---- C4
    import polytope_amd.polytope as pc
    from polytope_amd import prop2partition as p2p
    from polytope_amd import solvers
    old = solvers.default_solver
    solvers.default_solver = "hip"
    try:
        shape = (10, 10, 5, 2)
        cells, index = [], []
        for idx in itertools.product(*[range(n) for n in shape]):
            cells.append(pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]))
            index.append(idx)
        index = np.array(index)
        p2p.adjacency_matrix_dense(cells)
        t0 = time.perf_counter()
        adj = p2p.adjacency_matrix_dense(cells)
        t_adj = time.perf_counter() - t0
        want = (np.abs(index[:, None, :] - index[None, :, :]).max(axis=2) <= 1).astype(np.int8)
        c4 = {"workload": "1000-cell 10x10x5x2 grid, d=4 (BASELINE configs[3])",
              "adjacency_pairs": 499500, "adjacency_ms": t_adj * 1e3, "pair_lps_per_s": 499500 / t_adj,
              "adjacency_equals_grid_neighbourhood": bool(np.array_equal(adj, want))}
        gpath = os.path.join(ROOT, "tests", "golden", "g12_config4.npz")
        if os.path.exists(gpath):
            g = np.load(gpath, allow_pickle=False)
            Pp = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
            pc.region_diff(Pp.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
            t0 = time.perf_counter()
            D = pc.region_diff(Pp.copy(), pc.Region(cells[:500]), _order=g["c4_order"])
            t_diff = time.perf_counter() - t0
            c4.update(region_diff_ms=t_diff * 1e3, region_diff_pieces=len(D), reference_pieces=234,
                      reference_lps=int(g["c4_diff_nlp"]), lps_per_s_reference_count=int(g["c4_diff_nlp"]) / t_diff)
        out["C4_region"] = c4
    finally:
        solvers.default_solver = old
    # ---- fused reduce (64,16) x 5000
    B, m, d = 5000, 64, 16
    A, b = synth.random_hpolytopes(B, m, d, seed=2)
    At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
    res = pa.reduce_batch(At, bt)
    ms = timeit(lambda: pa.reduce_batch(At, bt), reps=5)
    nlp = int(res["nlp"].sum().item())
    by = B * (8 * m * (d + 1) + 12)
    keep, flags, nl, r = (res[k].cpu().numpy() for k in ("keep", "flags", "nlp", "r"))
    ok = True
    for k in range(48):                            # 48 polytopes (3 400 LPs) on the oracle
        o = O.reduce(A[k], b[k])
        ok = ok and (int(keep[k]) & (2 ** 64 - 1)) == int(o["mask"]) and int(flags[k]) == o["flags"] and int(nl[k]) == o["nlp"] \
            and abs(r[k] - o["r"]) <= 1e-9
    out["reduce_64x16"] = {"workload": "fused reduce of 5000 random H-polytopes, m=64, d=16", "ms": ms, "lps": nlp,
                           "lp_per_s": nlp / (ms * 1e-3),
                           "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "parity": {"oracle_polytopes": 48, "equal": bool(ok)}}
    return out

def _oracle_chunk(args):
    from oracle import oracle as O
    lo, cnt = args
    A, b = _ORACLE_AB
    lps = 0
    for k in range(lo, min(lo + cnt, A.shape[0])):
        lps += O.reduce(A[k], b[k])["nlp"]
    return lps


def _scipy_chunk(args):
    """the F2 LPs of a few polytopes on one worker process; BLAS / OpenMP pools held to one thread (one process per core)"""
    from scipy.optimize import linprog
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except Exception:
        lim = None
    cnt = 0
    for Ak, bk in args:
        for row in range(Ak.shape[0]):
            h = bk.copy()
            h[row] += 0.1
            linprog(-Ak[row], Ak, h, None, None, bounds=(None, None))
            cnt += 1
    if lim is not None:
        lim.restore_original_limits()
    return cnt


# ---- full-batch parity (outside the timed region): every polytope of every bench batch against the oracle -------------
_PAR = {}


def _parity_task(args):
    """Pool worker (forked before HIP / RCCL exist in the parent): regenerates batch (seed, stream) -- Philox is
    counter-based, so this is the rank's own data -- and runs the oracle's reduce() on polytopes [lo, hi)."""
    import importlib.util
    seed, stream, lo, hi = args
    if _PAR.get("key") != (seed, stream):
        if "synth" not in _PAR:   # by path: importing the package would load libplp_hip.so into every worker
            spec = importlib.util.spec_from_file_location("_plp_synth", os.path.join(ROOT, "polytope_amd", "synth.py"))
            _PAR["synth"] = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(_PAR["synth"])
        _PAR["AB"] = _PAR["synth"].random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=seed, stream=stream)
        _PAR["key"] = (seed, stream)
    from oracle import oracle as O
    A, b = _PAR["AB"]
    R = O.reduce_batch(A[lo:hi], b[lo:hi])
    return seed, lo, hi, R["keep"], R["flags"], R["nlp"], R["r"]


def parity_check(pool, nproc, gpu_results, stream, r_tol=1e-9):
    """gpu_results[i] = dict(keep, flags, nlp, r) (numpy) of batch i (seed i, this rank's stream).  Every polytope is
    reduced again by oracle/plp_oracle.c (polytope.py:1053-1163 restated) on the host cores: keep mask, flags and LP
    count must be equal, the Chebyshev radius within r_tol (north_star: 1e-9)."""
    import numpy as np
    t0 = time.perf_counter()
    NB = len(gpu_results)
    per_batch = max(1, -(-nproc // NB))
    step = -(-B_PER_GPU // per_batch)
    tasks = [(i, stream, lo, min(B_PER_GPU, lo + step)) for i in range(NB) for lo in range(0, B_PER_GPU, step)]
    bad = {"keep": 0, "flags": 0, "nlp": 0, "r": 0}
    max_r, checked, lps = 0.0, 0, 0
    for seed, lo, hi, keep, flags, nlp, r in pool.imap_unordered(_parity_task, tasks, chunksize=1):
        g = gpu_results[seed]
        bad["keep"] += int(np.count_nonzero(g["keep"][lo:hi].view(np.uint64) != keep))
        bad["flags"] += int(np.count_nonzero(g["flags"][lo:hi] != flags))
        bad["nlp"] += int(np.count_nonzero(g["nlp"][lo:hi] != nlp))
        dr = np.abs(g["r"][lo:hi] - r)
        bad["r"] += int(np.count_nonzero(~(dr <= r_tol)))
        max_r = max(max_r, float(dr.max()))
        checked += hi - lo
        lps += int(nlp.sum())
    return {"checked": checked, "ok": not any(bad.values()), "mismatches": bad, "max_abs_r_err": max_r, "r_tol": r_tol,
            "oracle_lps": lps, "seconds": time.perf_counter() - t0, "processes": nproc,
            "what": "every polytope of every bench batch: keep mask / flags / LP count equal to oracle/plp_oracle.c's "
                    "reduce(), Chebyshev radius within r_tol; outside the timed region"}


def spawn_ranks(n):
    """`python bench.py --gpus N` started without a launcher: start the N ranks here (the environment
    torch.distributed.run would give them: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rendezvous on 127.0.0.1), rank r on
    cuda:r.  Rank 0 inherits stdout (the ONE JSON line), the others write to stderr.  Returns the first non-zero exit
    code; a rank that fails takes the others down (exact PIDs)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on these hosts
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c
                sys.stderr.write("bench.py: rank %d exited with %d, stopping the other ranks\n" % (r, c))
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.2 ms each
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-batch parity check against the oracle")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the `secondary` object (BASELINE configs 3-5 and the "
                    "(64,16) fused reduce, about three seconds)")
    ap.add_argument("--batches", type=int, default=6, help="distinct 100k-polytope batches the steps rotate over "
                    "(6 x 52.4 MB > 256 MiB Infinity Cache: every step reads its input from HBM)")
    ap.add_argument("--regions", type=int, default=5, help="timed regions (each: barrier + synchronize, the steps, "
                    "synchronize + barrier; max over ranks); the line reports the median region")
    ap.add_argument("--min-region-ms", type=float, default=50.0, help="a region is the K steps repeated until it lasts "
                    "at least this long (a 20-step region is 4 ms: the figure would move with the box's noise)")
    ap.add_argument("--force-dist", action="store_true", help="take the N > 1 code path (process group, exchange "
                    "buffers, overlapped all-gather) even at world size 1: a plumbing check of the RCCL path on a 1-GPU box")
    ap.add_argument("--pipelined", action="store_true", help="N = 1: after the timed regions run the same steps again "
                    "with two batches in flight (two HIP streams) and report them as the extra object `pipelined`. "
                    "Off by default so that a rocprofv3 run of the default command sees single-launch dispatches only")
    ap.add_argument("--streams", type=int, default=1, help="issue the steps round-robin on this many HIP streams "
                    "(independent batches in flight: the half-empty last round of one launch overlaps the next "
                    "launch; with N > 1 the exchange is ordered against them at group boundaries).  Default 1: one "
                    "launch at a time, the regime roofline.kernel_ms and the rocprofv3 per-kernel durations describe; "
                    "`--pipelined` reports the two-stream throughput beside it")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak (default): `value` = every GPU "
                    "reduces its own 100k-polytope batches; at N > 1 the line also carries the object `strong` (ONE "
                    "100k-polytope batch per step partitioned across the GPUs in contiguous shards, north_star, "
                    "reassembled on every rank by the all-gather) measured in the same run.  strong: only that, as `value`")
    ap.add_argument("--gather-every", type=int, default=8, help="N > 1: batches per all-gather (G x 2.4 MB per rank)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "exercise the N>1 code path with several ranks on one GPU)")
    ap.add_argument("--verify-exchange", action="store_true", help="N > 1, after the timed regions: every rank recomputes "
                    "every rank's results of the last group of batches and compares them with what the all-gather "
                    "delivered (keep / flags / nlp / r bits); the line gets `exchange_verified`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be one.  (Under torch.distributed.run WORLD_SIZE is set and this process IS a rank.)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start the ranks with --nproc-per-node %d (or without a launcher: "
                         "bench.py spawns them itself)" % (args.gpus, world, args.gpus))

    # The parity pool is forked NOW, before HIP and RCCL exist in this process (forking a process that holds a HIP
    # context and RCCL's threads is asking for trouble); it idles until the timed regions are over.
    pool, pool_n = None, 0
    if not args.no_parity:
        import multiprocessing as mp
        from oracle import oracle as O
        O.build()   # once, here: the workers only load it
        pool_n = max(1, (os.cpu_count() or 1) // world)
        pool = mp.get_context("fork").Pool(pool_n)

    import torch
    import polytope_amd as pa
    from polytope_amd import _lib
    from polytope_amd.dist import GroupedExchange, shard_bounds
    from polytope_amd.synth import random_hpolytopes

    if not torch.cuda.is_available() or not _lib.available():
        raise SystemExit("bench.py needs a MI355X and polytope_amd/libplp_hip.so (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        raise SystemExit("--gpus %d over RCCL needs %d visible GPUs, this box has %d (several ranks on one GPU: "
                         "--backend gloo, a plumbing check only)" % (world, world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    rdev = dev if (multi and args.backend == "nccl") else torch.device("cpu")

    def allred(val, dtype, op=None):
        if not multi:
            return val
        t = torch.tensor([val], dtype=dtype, device=rdev)
        dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM)
        return t.item()

    NB = max(1, args.batches)
    G = max(1, args.gather_every)
    K = max(1, args.steps)
    counting = os.environ.get("PLP_BENCH_NO_COUNT", "0") != "1"
    main_st = torch.cuda.current_stream()

    def run_mode(dev_batches, B_LOCAL):
        """One scaling mode on this rank's resident batches: warm-up, a calibration region, then `--regions` timed
        regions of the same steps (every region starts the rotation at batch 0, so the regions are the same work)."""
        nb = 24 * B_LOCAL
        # N > 1: the kernel writes its results straight into a slot of a flat exchange buffer (24 B per polytope, no
        # packing kernels); every G batches the buffer goes out as ONE all-gather (xGMI is point-to-point: fewer,
        # larger collectives), on RCCL's stream while the next group of batches is computed in the other buffer.
        # Every batch's results are on every rank before a timed region ends (drain below).
        ex = GroupedExchange(torch, dist, B_LOCAL, DIM, G, dev) if multi else None
        # --streams S > 1: the steps are issued round-robin on S HIP streams (S independent batches in flight); the
        # exchange (N > 1) is ordered against the side streams at group boundaries only.
        side = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
        nissued = [0]

        def join():
            if side is not None:
                for st in side:
                    main_st.wait_stream(st)

        def step():
            k = nissued[0]
            nissued[0] += 1
            At, bt = dev_batches[k % NB]
            if side is None:
                if ex is None:
                    return pa.reduce_batch(At, bt)  # the fused kernel on torch's current stream
                res = pa.reduce_batch(At, bt, out=ex.slot().views)
                ex.commit()
                return res
            st = side[k % len(side)]
            if ex is None:
                with torch.cuda.stream(st):
                    return pa.reduce_batch(At, bt)
            if ex.k % G == 0:  # a new group starts in the other buffer: its last all-gather was waited for on `main`
                for s_ in side:
                    s_.wait_stream(main_st)
            with torch.cuda.stream(st):
                res = pa.reduce_batch(At, bt, out=ex.slot().views)
            if ex.k % G == G - 1:  # the group is complete: its kernels must have run before the all-gather reads it
                join()
            ex.commit()
            return res

        def drain():
            join()
            return ex.drain()

        # every batch once, untimed: loads the code object and yields the results the parity check compares and the LP
        # count of each batch (the number of lpsolve() calls the reference would issue on it: kernel output nlp[])
        first = [pa.reduce_batch(At_, bt_) for At_, bt_ in dev_batches]
        nlp_of = [int(v["nlp"].sum().item()) for v in first]

        def region(nsteps):
            """barrier + synchronize | nsteps steps (+ the drain of the exchange) | synchronize + barrier; the wall time
            is the max over ranks.  One HIP event pair sits around the launches on the stream they are issued on."""
            nissued[0] = 0
            if ex is not None:
                ex.k = 0
            torch.cuda.synchronize()
            if counting:
                pa.batch.reduce_simplex_runs(dev_index, reset=True)
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            evs = side[0] if side is not None else main_st
            n_on_evs = (nsteps + len(side) - 1) // len(side) if side is not None else nsteps
            gathered = None
            t0 = time.perf_counter()
            if side is not None:
                for st in side:
                    st.wait_stream(main_st)
            ev0.record(evs)
            for _ in range(nsteps):
                res = step()
            ev1.record(evs)
            join()
            if ex is not None:
                gathered = drain()[-1]
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            el = float(allred(el, torch.float64, dist.ReduceOp.MAX if multi else None))
            simplex = pa.batch.reduce_simplex_runs(dev_index) if counting else None
            assert int(res["nlp"].sum().item()) == nlp_of[(nsteps - 1) % NB]   # the last step's own output
            if multi:
                assert gathered.numel() == world * G * nb
            return {"elapsed": el, "kern_ms": ev0.elapsed_time(ev1) / n_on_evs, "simplex": simplex, "steps": nsteps,
                    "gathered": gathered}

        for _ in range(args.warmup):
            step()
        if ex is not None:
            drain()
        cal = region(K)["elapsed"]
        # (the calibration region is the first region of a mode and runs a few percent slower than the ones that follow: 15 % on top)
        rep = max(1, int(-(-1.15 * args.min_region_ms * 1e-3 // cal))) if cal > 0 else 1
        S = K * rep
        regs = [region(S) for _ in range(max(1, args.regions))]
        order = sorted(range(len(regs)), key=lambda i: regs[i]["elapsed"])
        med = regs[order[len(order) // 2]]
        lps_local = sum(nlp_of[k % NB] for k in range(S))
        lps = int(allred(lps_local, torch.int64))
        simplex = med["simplex"]
        if simplex is not None:
            simplex = int(allred(simplex, torch.int64))
        return {"ex": ex, "first": first, "nlp_of": nlp_of, "S": S, "rep": rep, "regs": regs, "med": med, "lps": lps,
                "lps_local": lps_local, "simplex": simplex, "B_LOCAL": B_LOCAL, "last_gathered": regs[-1]["gathered"],
                "region_ms": [r_["elapsed"] * 1e3 for r_ in regs]}

    def verify_exchange(mode, strong):
        # the last gathered group holds the slots of the steps of the (possibly partly filled) last group; rank q's part
        # must be what a reduce of rank q's batch of that step returns -- recomputed here, on this rank's GPU
        ex, S = mode["ex"], mode["S"]
        gcpu = mode["last_gathered"].cpu()
        first = (S - 1) // G * G
        ok, nslots = True, 0
        for kstep in range(first, S):
            s_ = kstep - first
            for q in range(world):
                if strong:
                    Aq, bq = random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=kstep % NB, stream=0)
                    loq, hiq = shard_bounds(B_PER_GPU, q, world)
                    Aq, bq = Aq[loq:hiq], bq[loq:hiq]
                else:
                    Aq, bq = random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=kstep % NB, stream=q)
                want = pa.reduce_batch(torch.as_tensor(Aq).to(dev), torch.as_tensor(bq).to(dev))
                got = ex.slot_views(gcpu, q, s_)
                for key in ("keep", "flags", "nlp"):
                    ok = ok and bool(torch.equal(got[key].cpu(), want[key].cpu()))
                ok = ok and bool(torch.equal(got["r"].cpu().view(torch.int64), want["r"].cpu().view(torch.int64)))
                nslots += 1
            if strong:   # the reassembled global batch, in batch order
                Ag, bg_ = random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=kstep % NB, stream=0)
                wantg = pa.reduce_batch(torch.as_tensor(Ag).to(dev), torch.as_tensor(bg_).to(dev))
                gv = ex.global_views(gcpu, s_)
                ok = ok and bool(torch.equal(gv["keep"].cpu(), wantg["keep"].cpu())) and bool(torch.equal(gv["nlp"].cpu(), wantg["nlp"].cpu()))
        allok = int(allred(1 if ok else 0, torch.int64, dist.ReduceOp.MIN))
        return {"ranks": world, "ok": allok == 1, "slots_checked": nslots}

    def shard_alone(dev_batches):
        # the floor of strong scaling: one rank's shard as a launch of its own (no exchange), event-timed
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            pa.reduce_batch(*dev_batches[0])
        e0.record()
        for k in range(50):
            pa.reduce_batch(*dev_batches[k % NB])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50

    # ---- the batches: NB distinct ones per mode, resident in HBM before any clock starts ---------------------------------
    want_weak = args.scaling == "weak"
    want_strong = args.scaling == "strong" or (multi and want_weak)
    weak = strong = None
    host_batches = None
    if want_weak:
        host_batches = [random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=i, stream=rank) for i in range(NB)]
        dev_w = [(torch.as_tensor(A_).to(dev), torch.as_tensor(b_).to(dev)) for A_, b_ in host_batches]
        weak = run_mode(dev_w, B_PER_GPU)
    if want_strong:
        # every rank regenerates the same global batches (counter-based RNG, stream 0) and keeps its contiguous shard
        if B_PER_GPU % world:
            raise SystemExit("strong scaling: %d polytopes do not split evenly over %d ranks" % (B_PER_GPU, world))
        lo, hi = shard_bounds(B_PER_GPU, rank, world)
        glob = host_batches if (want_weak and rank == 0) else [
            random_hpolytopes(B_PER_GPU, M_ROWS, DIM, seed=i, stream=0) for i in range(NB)]
        if host_batches is None:
            host_batches = glob
        dev_s = [(torch.as_tensor(A_[lo:hi]).to(dev), torch.as_tensor(b_[lo:hi]).to(dev)) for A_, b_ in glob]
        strong = run_mode(dev_s, hi - lo)
        strong["shard_alone_ms"] = shard_alone(dev_s)
    head = weak if want_weak else strong      # the mode `value` is quoted on
    A, b = host_batches[0]

    # ---- after the clocks: self-check of the collective layer, exchange verification, full-batch parity -------------------
    ranks_seen, exchange_ms = None, None
    if multi:
        ranks_seen = int(allred(1, torch.int64))
        if ranks_seen != world:   # a collective layer that does not see every rank measures something else: refuse the number
            raise SystemExit("bench.py: the all-reduce over the process group counted %d ranks, %d were launched (backend %s)"
                             % (ranks_seen, world, args.backend))
        # what ONE exchange of a group costs on its own (all-gather of G x 24 B x polytopes per rank, nothing overlapping it)
        src = head["ex"].big[0]
        src = src.cpu() if args.backend == "gloo" else src
        dst = torch.empty((world * src.numel(),), dtype=torch.uint8, device=src.device)
        for _ in range(3):
            dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        dist.barrier()
        te = time.perf_counter()
        nrep = 10
        for _ in range(nrep):
            dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        exchange_ms = float(allred((time.perf_counter() - te) / nrep * 1e3, torch.float64, dist.ReduceOp.MAX))
    verified = None
    if multi and args.verify_exchange:
        verified = verify_exchange(head, strong=not want_weak)
        if want_weak and strong is not None:
            verified["strong"] = verify_exchange(strong, strong=True)
    parity = None
    if pool is not None:
        # this rank's own batches (weak: stream = rank; strong only: the global batches, whose shards it reduced --
        # checked as whole batches through a launch of their own, the same kernel on the same rows)
        if want_weak:
            gres = [{k: v[k].cpu().numpy() for k in ("keep", "flags", "nlp", "r")} for v in weak["first"]]
            stream = rank
        else:
            gres = []
            for A_, b_ in host_batches:
                v = pa.reduce_batch(torch.as_tensor(A_).to(dev), torch.as_tensor(b_).to(dev))
                gres.append({k: v[k].cpu().numpy() for k in ("keep", "flags", "nlp", "r")})
            stream = 0
        parity = parity_check(pool, pool_n, gres, stream)
        pool.close()
        pool.join()
        tot = int(allred(parity["checked"], torch.int64))
        allok = int(allred(1 if parity["ok"] else 0, torch.int64, dist.ReduceOp.MIN if multi else None))
        parity["checked_all_ranks"], parity["ok_all_ranks"] = tot, allok == 1

    if rank == 0:
        med = head["med"]
        B_LOCAL = head["B_LOCAL"]
        kern_ms = med["kern_ms"]
        elapsed, S = med["elapsed"], head["S"]
        alg_bytes = B_LOCAL * (8 * M_ROWS * (DIM + 1) + 12)  # 524 B per (16,3) polytope
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: they need their own counters-only rocprofv3
        # passes (scripts/gpu_check.sh), whose summary is committed by scripts/summarize_profiles.py
        traffic, traffic_src, valu = None, None, {}
        kname = "reduce_lane_mix_kernel<3, 16, 4, 8>"   # plp_reduce_lane.hip: tiles of 16 polytopes, the last eighth as tiles of 8
        try:
            with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
                tj = json.load(f)
            if tj.get("kernel") != kname:
                raise KeyError("counters of another kernel")
            traffic = tj["hbm_bytes_per_launch"]
            valu = {k: tj[k] for k in ("valu_insts_per_launch", "valu_busy_frac_measured", "busy_frac_basis", "effective_clock_GHz", "source")
                    if tj.get(k) is not None}
            traffic_src = "profiles/latest_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)"
        except Exception:
            pass
        simplex = head["simplex"]
        npoly = (B_PER_GPU * world if want_weak else B_PER_GPU) * S   # polytopes reduced per region, all ranks
        line = {
            "metric": "LP solves/sec (batched Chebyshev + redundancy)",
            "value": head["lps"] / elapsed,
            "unit": "LP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / S * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "polytopes_per_s": npoly / elapsed,
            "config": {"workload": "reduce() of %d random H-polytopes per GPU, d=%d, m=%d (BASELINE configs[1]); %d distinct "
                                   "batches resident in HBM, one per step in rotation (%.0f MB > 256 MiB Infinity Cache)"
                                   % (B_PER_GPU, DIM, M_ROWS, NB, NB * B_PER_GPU * 8 * M_ROWS * (DIM + 1) / 1e6),
                       "lps_per_step": head["lps"] / S, "batches": NB, "polytopes_per_gpu": B_LOCAL, "streams": args.streams,
                       "timing": "`--steps` K = %d; a timed region = K x %d = %d steps (>= %.0f ms), bracketed by barrier + "
                                 "synchronize, max over ranks; %d regions, the line reports the median one"
                                 % (K, head["rep"], S, args.min_region_ms, len(head["regs"])),
                       "timed_steps_per_region": S, "regions": len(head["regs"]), "region_ms": head["region_ms"],
                       "lp_accounting": "`value` counts the LPs the reference issues on these polytopes (kernel output nlp, checked "
                                        "against the oracle's count); `lps_simplex_per_step` of them ran the simplex in the timed "
                                        "steps (device counter, plp_reduce_counters), the others are redundancy LPs whose verdict the "
                                        "two-witness presolve settled (DESIGN.md 4.1): `presolved_frac` of all LPs",
                       "lps_simplex_per_step": None if simplex is None else simplex / S,
                       "presolved_frac": None if simplex is None else 1.0 - simplex / head["lps"],
                       "value_simplex_only": None if simplex is None else simplex / elapsed,
                       "parallelism": "batch-sharded x%d + all-gather of the packed results of every %d batches (overlapped with the next ones)" % (world, G)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "VALU-issue bound (valu_busy_frac_measured), not HBM bound: %.3g LP/s inside the kernel" % (
                             head["lps_local"] / S / (kern_ms * 1e-3))},
        }
        if parity is not None:
            line["parity_checked"] = parity["checked_all_ranks"]
            line["parity_ok"] = parity["ok_all_ranks"]
            line["parity"] = parity
        if multi:
            line["config"]["rccl_ranks_seen"] = ranks_seen
            line["config"]["backend"] = args.backend
            line["config"]["exchange_ms_per_group"] = exchange_ms
            line["config"]["exchange_bytes_per_rank_per_group"] = G * 24 * B_LOCAL
        if strong is not None:
            sm = strong["med"]
            floor = {"shard_polytopes": strong["B_LOCAL"], "shard_kernel_ms_alone": strong["shard_alone_ms"],
                     "note": "one rank's shard as a launch of its own, no exchange: what a step of the partitioned batch "
                             "cannot go below on this build"}
            sobj = {"what": "ONE %d-polytope batch per step partitioned over the %d GPUs in contiguous shards (north_star), "
                            "reassembled on every rank by the all-gather; same run, same batches as rank 0's weak ones"
                            % (B_PER_GPU, world),
                    "value": strong["lps"] / sm["elapsed"], "unit": "LP/s", "ms_per_step": sm["elapsed"] / strong["S"] * 1e3,
                    "polytopes_per_s": B_PER_GPU * strong["S"] / sm["elapsed"], "lps_per_step": strong["lps"] / strong["S"],
                    "timed_steps_per_region": strong["S"], "region_ms": strong["region_ms"],
                    "kernel_ms": sm["kern_ms"], "strong_floor": floor}
            if want_weak:
                line["strong"] = sobj
            line["config"]["strong_floor"] = floor
        if verified is not None:
            line["exchange_verified"] = verified
        if valu:  # measured PMC counters of the same kernel (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU over GRBM_GUI_ACTIVE)
            line["roofline"].update({k: v for k, v in valu.items() if k != "source"})
            line["roofline"]["counters_source"] = valu.get("source")
            if "valu_insts_per_launch" in valu:
                # the bound that matters: a VALU instruction of a wavefront occupies its SIMD for 4 cycles; 256 CUs x 4
                # SIMDs at 2.4 GHz (MI355X_MICROARCH.md).  Counter from the committed PMC pass, time from THIS run.
                line["roofline"]["valu_issue_frac"] = valu["valu_insts_per_launch"] * 4.0 / (
                    1024 * SIMD_CLOCK_HZ * kern_ms * 1e-3) * (B_LOCAL / B_PER_GPU)
                line["roofline"]["valu_issue_note"] = ("SQ_INSTS_VALU per launch (committed PMC pass) x 4 cycles / (1024 SIMDs x "
                                                       "2.4 GHz x kernel_ms of this run)")
        if args.pipelined and not multi and args.streams == 1:
            # Not `value`: the same steps again with two independent batches in flight (two HIP streams).  100 000
            # polytopes are 6250 wavefronts for 4096 resident slots, so the last round of a launch runs half empty;
            # with a second launch in flight that hole is filled -- what a caller with a stream of batches should do.
            two = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for k in range(2 * max(1, args.warmup)):  # untimed: the streams' queues are created on first use
                with torch.cuda.stream(two[k & 1]):
                    pa.reduce_batch(*dev_w[k % NB])
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for st in two:
                st.wait_stream(torch.cuda.current_stream())
            for k in range(S):
                with torch.cuda.stream(two[k & 1]):
                    res2 = pa.reduce_batch(*dev_w[k % NB])
            torch.cuda.synchronize()
            tp = time.perf_counter() - tp
            assert int(res2["nlp"].sum().item()) == head["nlp_of"][(S - 1) % NB]
            line["pipelined"] = {"streams": 2, "value": head["lps_local"] / tp, "unit": "LP/s",
                                 "ms_per_step": tp / S * 1e3,
                                 "note": "two batches in flight; not the headline, see DESIGN.md section 6"}
        if not multi and not args.no_end_to_end:
            line["end_to_end"] = end_to_end(torch, pa, A, b, dev)
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(A, b, head["first"][0]["nlp"].cpu().numpy())
        if not multi and not args.no_secondary:
            t_sec = time.perf_counter()
            line["secondary"] = secondary(torch, pa, dev)
            line["secondary"]["seconds"] = time.perf_counter() - t_sec
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    if parity is not None and not parity["ok_all_ranks"]:
        raise SystemExit("bench.py: the GPU results differ from the oracle's (see `parity` in the line)")


if __name__ == "__main__":
    main()
