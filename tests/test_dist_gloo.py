"""CPU: the N>1 path (contiguous sharding + one fused all-gather of packed results) under
torch.distributed gloo, world_size 2 and 3 (uneven shards).  The per-shard reduce is a
stand-in built on the oracle -- what is under test is polytope_amd.dist, not the kernel."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polytope_amd import dist as pdist
    from polytope_amd.synth import random_hpolytopes
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b = random_hpolytopes(B, 10, 2, seed=5)  # every rank regenerates the same batch (Philox)

    def cpu_reduce(As, bs, ms, tol):
        keep, flags, nlp, r = [], [], [], []
        for k in range(As.shape[0]):
            o = O.reduce(As[k], bs[k], tol)
            keep.append(np.int64(np.uint64(o["mask"]).astype(np.int64)))
            flags.append(o["flags"]); nlp.append(o["nlp"]); r.append(o["r"])
        return dict(keep=torch.tensor(np.array(keep, dtype=np.int64)), flags=torch.tensor(flags, dtype=torch.int32),
                    nlp=torch.tensor(nlp, dtype=torch.int32), r=torch.tensor(r, dtype=torch.float64))

    res = pdist.reduce_batch_sharded(A, b, reduce_fn=cpu_reduce)
    lo, hi = pdist.shard_bounds(B, rank, world)
    q.put((rank, lo, hi, res["keep"].numpy(), res["flags"].numpy(), res["nlp"].numpy(), res["r"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 64), (3, 50)])
def test_sharded_reduce_allgather(world, B):
    import torch.multiprocessing as mp
    from polytope_amd.synth import random_hpolytopes
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b = random_hpolytopes(B, 10, 2, seed=5)
    want = [O.reduce(A[k], b[k]) for k in range(B)]
    covered = np.zeros(B, bool)
    for rank, lo, hi, keep, flags, nlp, r in outs:
        covered[lo:hi] = True
        assert keep.shape == (B,)  # every rank holds the whole reassembled batch
        assert [int(x) for x in keep.astype(np.uint64)] == [w["mask"] for w in want]
        assert list(flags) == [w["flags"] for w in want]
        assert list(nlp) == [w["nlp"] for w in want]
        assert np.array_equal(r, np.array([w["r"] for w in want]))
    assert covered.all()


def _worker_points(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polytope_amd import dist as pdist
    from polytope_amd.synth import containment_workload, quickhull_workload
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, X = containment_workload(12, 1001, d=3, m=8, seed=2)

    def cpu_contains(A_, b_, Xs, tol, m_):
        return torch.as_tensor(O.contains(A_, b_, np.ascontiguousarray(Xs.T), abs_tol=tol, mrows=m_, region=True))

    got = pdist.contains_sharded(A, b, X, 1e-7, contains_fn=cpu_contains)
    Xq, nrm, off = quickhull_workload(777, d=3, F=9, seed=5)
    Xq[500] = Xq[100]  # exact tie across ranks: the lower global index must win

    def cpu_assign(Xs, n_, o_, tol):
        fo, dd, am, mx = O.assign(Xs, n_, o_, tol)
        return dict(facet=torch.as_tensor(fo), dist=torch.as_tensor(dd), argmax=torch.as_tensor(am),
                    maxd=torch.as_tensor(np.where(am >= 0, mx, 0.0)))

    res = pdist.assign_sharded(Xq, nrm, off, 1e-7, assign_fn=cpu_assign)
    q.put((rank, got.numpy(), res["facet"].numpy(), res["dist"].numpy(), res["argmax"].numpy(), res["maxd"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_contains_and_assign(world):
    import torch.multiprocessing as mp
    from polytope_amd.synth import containment_workload, quickhull_workload
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_points, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b, X = containment_workload(12, 1001, d=3, m=8, seed=2)
    want = O.contains(A, b, np.ascontiguousarray(X.T), abs_tol=1e-7, region=True)
    Xq, nrm, off = quickhull_workload(777, d=3, F=9, seed=5)
    Xq[500] = Xq[100]
    fo, dd, am, mx = O.assign(Xq, nrm, off, 1e-7)
    for rank, got, facet, dist_, argmax, maxd in outs:
        assert np.array_equal(got, want)
        assert np.array_equal(facet, fo) and np.array_equal(dist_, dd)
        assert np.array_equal(argmax, am), (rank, argmax, am)
        assert np.array_equal(maxd[am >= 0], mx[am >= 0])


def test_shard_bounds_and_packing():
    import torch
    from polytope_amd import dist as pdist
    for B in (0, 1, 7, 100000):
        for world in (1, 2, 3, 8):
            cuts = [pdist.shard_bounds(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    res = dict(keep=torch.tensor([-1, 5, 1 << 62], dtype=torch.int64), flags=torch.tensor([4, 1, 2], dtype=torch.int32),
               nlp=torch.tensor([23, 1, 7], dtype=torch.int32), r=torch.tensor([1.5, 0.0, 1e-9], dtype=torch.float64))
    back = pdist.unpack_results(torch, pdist.pack_results(torch, res))
    for k in res:
        assert torch.equal(back[k], res[k]), k


def _grid_cells(shape):
    """Unit boxes of a grid as stacked H-polytopes A[n, 2d, d], b[n, 2d]."""
    import itertools
    d = len(shape)
    lo = np.array(list(itertools.product(*[range(s) for s in shape])), dtype=float)
    A = np.tile(np.vstack([np.eye(d), -np.eye(d)]), (lo.shape[0], 1, 1))
    b = np.concatenate([lo + 1.0, -lo], axis=1)
    return lo, A, b


def _worker_adj_hull(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polytope_amd import dist as pdist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, A, b = _grid_cells((4, 3, 2))

    def cpu_pairs(A_, b_, p_lo, p_hi, m_, tol):  # oracle stand-in for plp_adjacent_pairs_range
        ii, jj = np.tril_indices(A_.shape[0], -1)
        out = np.zeros(p_hi - p_lo, np.uint8)
        for p in range(p_lo, p_hi):
            st, r, _ = O.cheby(np.vstack([A_[ii[p]], A_[jj[p]]]), np.r_[b_[ii[p]], b_[jj[p]]] + tol)
            out[p - p_lo] = 1 if (st == 0 and r > tol / 10) else 0
        return out

    # overlapped exchange: push k returns the gathered rows of push k-1
    pipe = pdist.GatherPipeline(torch, dist, rows=5, cols=3)
    got = []
    for k in range(4):
        prev = pipe.push(torch.full((5, 3), 100 * k + rank, dtype=torch.int64))
        got.append(None if prev is None else prev.clone())  # the buffer is reused two pushes later
    got.append(pipe.flush())
    assert got[0] is None
    for k in range(4):
        want = torch.cat([torch.full((5, 3), 100 * k + r, dtype=torch.int64) for r in range(world)])
        assert torch.equal(got[k + 1], want), (k, rank)
    # coalesced exchange (what bench.py --gpus N runs): G batches per all-gather, results written into slot views
    Bq, G = 7, 3
    ex = pdist.GroupedExchange(torch, dist, Bq, 3, G, None)
    groups = []
    for k in range(8):                                   # 2 full groups + a partly filled one
        v = ex.slot().views
        v["keep"][:] = 1000 * k + rank
        v["r"][:] = k + 0.25 * rank
        v["flags"][:] = k
        v["nlp"][:] = 10 * k + rank
        out = ex.commit()
        if out is not None:
            groups.append(out.clone())                   # an EARLIER group (buffers are reused two pushes later)
    groups += [g.clone() for g in ex.drain()]
    assert len(groups) == 3 and all(g.numel() == world * G * 24 * Bq for g in groups)
    for gi, g in enumerate(groups):
        for r in range(world):
            for sl in range(G):
                k = gi * G + sl
                if k >= 8:
                    continue                             # slots of the last group that were not filled
                w = ex.slot_views(g, r, sl)
                assert int(w["keep"][0]) == 1000 * k + r and int(w["keep"][-1]) == 1000 * k + r, (gi, r, sl)
                assert float(w["r"][3]) == k + 0.25 * r and int(w["flags"][2]) == k and int(w["nlp"][6]) == 10 * k + r
    adj = pdist.adjacent_pairs_sharded(A, b, pairs_fn=cpu_pairs).numpy()
    P = np.random.default_rng(3).standard_normal((3000, 3))
    np.random.seed(5)   # every rank draws the same start simplex
    Ah, bh, Vh = pdist.quickhull_sharded(P, session_factory=O.HullSession)
    q.put((rank, adj, Ah, bh, Vh))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adjacency_and_quickhull(world):
    """Pair-space sharding of find_adjacent_regions and point sharding of quickhull's outside sets:
    every rank ends with the full adjacency matrix / the same hull as the single-process run."""
    import torch.multiprocessing as mp
    from scipy.spatial import ConvexHull
    from polytope_amd.quickhull import quickhull
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_adj_hull, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lo, A, b = _grid_cells((4, 3, 2))
    touching = np.all(np.abs(lo[:, None, :] - lo[None, :, :]) <= 1.0, axis=2)
    P = np.random.default_rng(3).standard_normal((3000, 3))
    np.random.seed(5)
    A1, b1, V1 = quickhull(P, session_factory=O.HullSession)   # single process, same session type
    ref = P[np.unique(ConvexHull(P).vertices)]
    ref = ref[np.lexsort(ref.T[::-1])]
    assert np.array_equal(V1, ref)
    for rank, adj, Ah, bh, Vh in outs:
        assert np.array_equal(adj.astype(bool), touching)
        assert np.array_equal(Ah, A1) and np.array_equal(bh, b1) and np.array_equal(Vh, V1)


def _worker_strong(rank, world, port, q):
    """bench.py --scaling strong: ONE batch per step, partitioned into equal contiguous shards; every rank reduces its
    shard straight into its slot of the exchange buffer and the coalesced all-gather reassembles the batch."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from polytope_amd import dist as pdist
    from polytope_amd.synth import random_hpolytopes
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, G, steps = 12 * world, 2, 5
    lo, hi = pdist.shard_bounds(B, rank, world)
    ex = pdist.GroupedExchange(torch, dist, hi - lo, 2, G, None)
    groups = []
    for k in range(steps):
        A, b = random_hpolytopes(B, 10, 2, seed=100 + k)   # the same global batch on every rank (Philox)
        v = ex.slot().views
        for t, p in enumerate(range(lo, hi)):
            o = O.reduce(A[p], b[p])
            v["keep"][t] = int(np.uint64(o["mask"]).astype(np.int64))
            v["flags"][t], v["nlp"][t], v["r"][t] = o["flags"], o["nlp"], o["r"]
        out = ex.commit()
        if out is not None:
            groups.append(out.clone())
    groups += [g.clone() for g in ex.drain()]
    res = []
    for k in range(steps):
        gv = ex.global_views(groups[k // G], k % G)
        res.append((gv["keep"].numpy().copy(), gv["flags"].numpy().copy(), gv["nlp"].numpy().copy(), gv["r"].numpy().copy()))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_strong_scaling_exchange(world):
    import torch.multiprocessing as mp
    from polytope_amd.synth import random_hpolytopes
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_strong, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B = 12 * world
    for k in range(5):
        A, b = random_hpolytopes(B, 10, 2, seed=100 + k)
        want = [O.reduce(A[p], b[p]) for p in range(B)]
        for rank, res in outs:   # every rank ends up with the whole batch of every step, in batch order
            keep, flags, nlp, r = res[k]
            assert [int(x) for x in keep.astype(np.uint64)] == [w["mask"] for w in want], (k, rank)
            assert list(flags) == [w["flags"] for w in want] and list(nlp) == [w["nlp"] for w in want]
            assert np.array_equal(r, np.array([w["r"] for w in want]))
