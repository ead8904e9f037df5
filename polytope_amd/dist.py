"""Multi-GPU sharding of the batched path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The LPs of different polytopes are independent, so the batch is split into contiguous
chunks (no data-path collective) and the only exchange step is ONE all-gather of the
packed per-polytope results (keep mask, flags, LP count, Chebyshev radius = 24 B per
polytope) that reassembles the reduced Region on every rank (north_star).  At B = 100k
that is 2.4 MB per rank: latency-bound on xGMI, so a single fused all-gather of one packed
buffer is used rather than one collective per array.
"""
import numpy as np


def shard_bounds(B, rank, world):
    """Contiguous chunk [lo, hi) of a batch of B items owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_results(torch, res):
    """keep/flags/nlp/r of reduce_batch -> one int64 tensor [B, 3] (flags and nlp share a word)."""
    keep = res["keep"].to(torch.int64)
    fl = res["flags"].to(torch.int64)
    nl = res["nlp"].to(torch.int64)
    r_bits = res["r"].contiguous().view(torch.int64)
    return torch.stack([keep, (nl << 32) | fl, r_bits], dim=1).contiguous()


def unpack_results(torch, packed):
    keep = packed[:, 0].contiguous()
    flags = (packed[:, 1] & 0xFFFFFFFF).to(torch.int32)
    nlp = (packed[:, 1] >> 32).to(torch.int32)
    r = packed[:, 2].contiguous().view(torch.float64)
    return dict(keep=keep, flags=flags, nlp=nlp, r=r)


def allgather_packed(torch, dist, packed, counts):
    """All-gather row blocks of possibly different sizes (counts[rank] rows each).

    Equal counts -> one all_gather_into_tensor; otherwise padded to the maximum.
    """
    world = dist.get_world_size()
    if world == 1:
        return packed
    cmax = max(counts)
    cols = packed.shape[1]
    if packed.shape[0] != cmax:
        pad = torch.zeros((cmax - packed.shape[0], cols), dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad], dim=0)
    if dist.get_backend() == "gloo" and packed.is_cuda:  # test-only path: gloo gathers on the host
        host = packed.contiguous().cpu()
        out = torch.empty((world * cmax, cols), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        out = out.to(packed.device)
    else:
        out = torch.empty((world * cmax, cols), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(out, packed.contiguous())
    if all(c == cmax for c in counts):
        return out
    parts = [out[r * cmax: r * cmax + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


class ResultBuffer:
    """One flat exchange buffer per batch: [keep int64 | r float64 | flags int32 | nlp int32] x B = 24 B per
    polytope, with typed views that reduce_batch(..., out=...) writes straight into, so that the all-gather
    needs no packing kernels.  `split(flat)` gives the same views on a gathered [world * nbytes] buffer."""

    def __init__(self, torch, B, d, device, flat=None):
        """`flat`: use this uint8[24 B] slice as storage (one slot of a larger buffer that is exchanged as a whole:
        fewer, larger collectives)."""
        self.torch, self.B = torch, int(B)
        self.nbytes = 24 * self.B
        self.flat = torch.empty((self.nbytes,), dtype=torch.uint8, device=device) if flat is None else flat
        assert self.flat.numel() == self.nbytes and self.flat.dtype == torch.uint8
        self.views = self._views(self.flat)
        self.views["xc"] = torch.empty((self.B, d), dtype=torch.float64, device=device)  # not exchanged

    def _views(self, flat):
        t, B = self.torch, self.B
        return dict(keep=flat[0:8 * B].view(t.int64), r=flat[8 * B:16 * B].view(t.float64),
                    flags=flat[16 * B:20 * B].view(t.int32), nlp=flat[20 * B:24 * B].view(t.int32))

    def split(self, gathered):
        """gathered uint8[world * nbytes] -> list (one per rank) of dict(keep, r, flags, nlp)"""
        world = gathered.numel() // self.nbytes
        return [self._views(gathered[k * self.nbytes:(k + 1) * self.nbytes]) for k in range(world)]


class GatherPipeline:
    """Overlap the exchange step with the next batch: the all-gather of batch k (RCCL's stream) runs while
    the reduce kernel of batch k+1 runs on the compute stream.  xGMI is point-to-point: an 8-rank ring
    all-gather of 8 x 2.4 MB costs a good fraction of the 0.28 ms kernel, so it is taken off the
    critical path instead of being paid after every kernel.

        pipe = GatherPipeline(torch, dist, rows=B, cols=3)        # or rows=nbytes, cols=1, dtype=torch.uint8
        for batch in batches:
            res = reduce_batch(...)                 # compute stream
            gathered_prev = pipe.push(pack_results(torch, res))   # results of the PREVIOUS batch (or None);
                                                                  # the buffer is reused two pushes later
        gathered_last = pipe.flush()
    """

    def __init__(self, torch, dist, rows, cols, dtype=None, device=None):
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size()
        dtype = dtype or torch.int64
        self.host = dist.get_backend() == "gloo"
        dev = torch.device("cpu") if self.host else device
        self.out = [torch.empty((self.world * rows, cols), dtype=dtype, device=dev) for _ in range(2)]
        self.k = 0
        self.work = None
        self.ready = None

    def push(self, packed):
        prev = self._wait()
        buf = self.out[self.k & 1]
        self.k += 1
        src = packed.contiguous().cpu() if self.host else packed.contiguous()
        self.keep = src  # alive until the collective has run
        self.work = self.dist.all_gather_into_tensor(buf, src, async_op=True)
        self.ready = buf
        return prev

    def _wait(self):
        if self.work is None:
            return None
        self.work.wait()  # the compute stream waits here; kernels enqueued before this line overlap the gather
        self.work = None
        return self.ready

    def flush(self):
        return self._wait()


class GroupedExchange:
    """The exchange step of a stream of batches, coalesced: the reduce kernel of batch k writes its results
    straight into slot k mod G of a flat buffer (ResultBuffer views, 24 B per polytope, no packing kernels) and
    every G batches the whole buffer goes out as ONE all-gather (xGMI is point-to-point: fewer, larger
    collectives) on RCCL's stream, while the next G batches are computed into the second buffer
    (GatherPipeline).  Every batch's results reach every rank; a partly filled group is exchanged by drain().

        ex = GroupedExchange(torch, dist, B, d, G, device)
        for batch in batches:
            res = reduce_batch(A, b, out=ex.slot().views)   # compute stream
            gathered = ex.commit()      # None, or the uint8[world * G * 24 B] buffer of an EARLIER group
        rest = ex.drain()               # the groups not handed out yet, oldest first (the last may be partly filled)
        ex.slot_views(gathered, rank, s)  # dict(keep, r, flags, nlp) of rank `rank`, slot s
    """

    def __init__(self, torch, dist, B, d, G, device):
        self.torch, self.G, self.nb = torch, int(G), 24 * int(B)
        host = dist.get_backend() == "gloo"
        dev = torch.device("cpu") if host and device is None else device
        self.big = [torch.zeros((self.G * self.nb,), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.bufs = [[ResultBuffer(torch, B, d, dev, flat=self.big[g][s * self.nb:(s + 1) * self.nb])
                      for s in range(self.G)] for g in range(2)]
        self.pipe = GatherPipeline(torch, dist, self.G * self.nb, 1, dtype=torch.uint8, device=dev)
        self.k = 0

    def slot(self):
        """ResultBuffer the next batch writes into"""
        return self.bufs[(self.k // self.G) & 1][self.k % self.G]

    def commit(self):
        """the batch written into slot() is complete (enqueued on the compute stream)"""
        k = self.k
        self.k += 1
        if k % self.G == self.G - 1:
            return self.pipe.push(self.big[(k // self.G) & 1].view(-1, 1))
        return None

    def drain(self):
        """Exchange a partly filled group and wait for what is in flight: the list of gathered groups that
        commit() has not handed out yet, oldest first (at most two; they live in different buffers)."""
        outs = []
        if self.k % self.G:
            prev = self.pipe.push(self.big[(self.k // self.G) & 1].view(-1, 1))
            if prev is not None:
                outs.append(prev.view(-1))
            self.k += self.G - self.k % self.G
        last = self.pipe.flush()
        if last is not None:
            outs.append(last.view(-1))
        return outs

    def slot_views(self, gathered, rank, s):
        lo = rank * self.G * self.nb + s * self.nb
        return self.bufs[0][0].split(gathered.view(-1)[lo:lo + self.nb])[0]

    def global_views(self, gathered, s):
        """Strong scaling (ONE batch partitioned into equal contiguous shards, rank r holding shard r): the results
        of the whole batch of slot s, reassembled in batch order from a gathered group -- dict(keep, r, flags, nlp)."""
        world = gathered.numel() // (self.G * self.nb)
        parts = [self.slot_views(gathered, r, s) for r in range(world)]
        return {k: self.torch.cat([p[k] for p in parts]) for k in parts[0]}


def reduce_batch_sharded(A, b, m=None, abs_tol=1e-7, reduce_fn=None, device=None):
    """Every rank holds (or can regenerate) the full batch A[B,m,d], b[B,m]; each rank reduces
    its contiguous shard and all ranks end up with the results of the whole batch.

    `reduce_fn(A_shard, b_shard, m_shard, abs_tol) -> dict of torch tensors` defaults to the HIP
    path (batch.reduce_batch on CUDA tensors); the CPU tests inject a stand-in so that the
    sharding / packing / all-gather logic runs under gloo without a GPU.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = A.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    counts = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    if reduce_fn is None:
        from .batch import reduce_batch
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

        def reduce_fn(As, bs, ms, tol):
            At = torch.as_tensor(np.ascontiguousarray(As)).to(dev)
            bt = torch.as_tensor(np.ascontiguousarray(bs)).to(dev)
            mt = None if ms is None else torch.as_tensor(np.ascontiguousarray(ms, dtype=np.int32)).to(dev)
            return reduce_batch(At, bt, mt, tol)
    res = reduce_fn(A[lo:hi], b[lo:hi], None if m is None else m[lo:hi], abs_tol)
    packed = pack_results(torch, res)
    if world > 1:
        packed = allgather_packed(torch, dist, packed, counts)
    return unpack_results(torch, packed)


# ------------------------------------------------------------------------------------------
# Containment: every rank holds all polytopes (a few MB, replicated) and a
# contiguous slice of the points; the exchange step is one all-gather of the uint8 results.
def contains_sharded(A, b, X, abs_tol=1e-7, m=None, contains_fn=None, device=None):
    """X[d, N] column vectors, known to every rank -> uint8[N] on every rank (Region.contains)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    N = X.shape[1]
    lo, hi = shard_bounds(N, rank, world)
    counts = [shard_bounds(N, r, world)[1] - shard_bounds(N, r, world)[0] for r in range(world)]
    if contains_fn is None:
        from .batch import contains_batch
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

        def contains_fn(A_, b_, Xs, tol, m_):
            t = lambda v, dt=None: None if v is None else torch.as_tensor(np.ascontiguousarray(v, dtype=dt)).to(dev)
            return contains_batch(t(A_), t(b_), t(Xs), tol, m=t(m_, np.int32), region=True)
    mine = contains_fn(A, b, np.ascontiguousarray(X[:, lo:hi]), abs_tol, m)
    mine = mine.to(torch.uint8).reshape(-1, 1)
    if world > 1:
        mine = allgather_packed(torch, dist, mine, counts)
    return mine.reshape(-1)


# Quickhull outside-set assignment / furthest point: shard the points; the
# per-point results stay with (or are gathered from) their owner, the per-facet furthest point needs
# one exchange: all-gather of F x (max distance, global index), then "first maximum wins" = the
# lowest global index among the ranks that attain the maximum (quickhull.py:97-100).
def assign_sharded(X, normals, offsets, abs_tol=1e-7, assign_fn=None, device=None, gather_points=True):
    """X[N, d] rows, known to every rank -> dict(facet[N], dist[N], argmax[F], maxd[F]) on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    N = X.shape[0]
    lo, hi = shard_bounds(N, rank, world)
    counts = [shard_bounds(N, r, world)[1] - shard_bounds(N, r, world)[0] for r in range(world)]
    if assign_fn is None:
        from .batch import assign_batch
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

        def assign_fn(Xs, nrm, off, tol):
            t = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to(dev)
            return assign_batch(t(Xs), t(nrm), t(off), tol)
    res = assign_fn(np.ascontiguousarray(X[lo:hi]), normals, offsets, abs_tol)
    am = res["argmax"].to(torch.int64)
    mx = res["maxd"].to(torch.float64)
    big = torch.iinfo(torch.int64).max
    gidx = torch.where(am >= 0, am + lo, torch.full_like(am, big))   # global point index, "none" = +inf
    mxv = torch.where(am >= 0, mx, torch.full_like(mx, -1.0))          # distances are > abs_tol >= 0
    if world > 1:
        pack = torch.stack([mxv.contiguous().view(torch.int64), gidx], dim=1).contiguous()   # [F, 2]
        F = pack.shape[0]
        allp = allgather_packed(torch, dist, pack, [F] * world).reshape(world, F, 2)
        dd = allp[:, :, 0].contiguous().view(torch.float64)            # [world, F]
        ii = allp[:, :, 1]
        best = dd.max(dim=0).values
        cand = torch.where(dd == best[None, :], ii, torch.full_like(ii, big))
        gidx = cand.min(dim=0).values
        mxv = best
    argmax = torch.where(gidx == big, torch.full_like(gidx, -1), gidx)
    maxd = torch.where(gidx == big, torch.zeros_like(mxv), mxv)
    out = dict(argmax=argmax, maxd=maxd)
    if gather_points:
        fac = res["facet"].to(torch.int64).reshape(-1, 1)
        dst = res["dist"].to(torch.float64).contiguous().view(torch.int64).reshape(-1, 1)
        both = torch.cat([fac, dst], dim=1).contiguous()
        if world > 1:
            both = allgather_packed(torch, dist, both, counts)
        out["facet"] = both[:, 0].to(torch.int32)
        out["dist"] = both[:, 1].contiguous().view(torch.float64)
    else:
        out["facet"], out["dist"], out["lo"], out["hi"] = res["facet"], res["dist"], lo, hi
    return out


# ------------------------------------------------------------------------------------------
# Adjacency of a partition: the n(n-1)/2 pair LPs are independent, so the pair
# index space p = i (i - 1) / 2 + j is split into contiguous slices, each rank solves its slice from the
# replicated cells (n x m x (d+1) doubles: tens of KB) and ONE all-gather of the per-pair bytes
# reassembles the matrix everywhere (499 500 B at n = 1000).
def adjacent_pairs_sharded(A, b, m=None, abs_tol=1e-7, pairs_fn=None, device=None):
    """A[n, m_max, d], b[n, m_max] known to every rank -> uint8[n, n] adjacency on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = A.shape[0]
    npairs = n * (n - 1) // 2
    lo, hi = shard_bounds(npairs, rank, world)
    counts = [shard_bounds(npairs, r, world)[1] - shard_bounds(npairs, r, world)[0] for r in range(world)]
    if pairs_fn is None:
        from .batch import adjacent_pairs_range
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

        def pairs_fn(A_, b_, lo_, hi_, m_, tol):
            t = lambda v, dt=None: None if v is None else torch.as_tensor(np.ascontiguousarray(v, dtype=dt)).to(dev)
            return adjacent_pairs_range(t(A_), t(b_), lo_, hi_, m=t(m_, np.int32), abs_tol=tol)
    mine = pairs_fn(A, b, lo, hi, m, abs_tol)
    mine = torch.as_tensor(mine).to(torch.uint8).reshape(-1, 1)
    if world > 1:
        mine = allgather_packed(torch, dist, mine, counts)
    flat = mine.reshape(-1)
    adj = torch.eye(n, dtype=torch.uint8, device=flat.device)
    ii, jj = np.tril_indices(n, -1)            # row-major lower triangle = the order p = i (i - 1) / 2 + j
    ii = torch.as_tensor(ii, device=flat.device)
    jj = torch.as_tensor(jj, device=flat.device)
    adj[ii, jj] = flat
    adj[jj, ii] = flat
    return adj


# ------------------------------------------------------------------------------------------
# Quickhull main loop with the outside sets sharded by points: every rank keeps a
# contiguous slice of the points resident on its GPU (a HullSession) and runs the SAME host facet graph;
# per iteration the one exchange step is an all-gather of 3 words per new facet (count, furthest distance,
# global index of the furthest point), combined as "largest distance, lowest global index on ties"
# (quickhull.py:97-100) and summed counts.
class ShardedHullSession:
    """Interface of polytope_amd.batch.HullSession over the points X[N, d] known to every rank."""

    def __init__(self, X, session_factory=None):
        import torch.distributed as dist
        self._dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        X = np.ascontiguousarray(X, dtype=float)
        self.N, self.d = X.shape
        self.lo, self.hi = shard_bounds(self.N, self.rank, self.world)
        if session_factory is None:
            from .batch import HullSession as session_factory
        self.local = session_factory(X[self.lo:self.hi])

    def drop(self, idx):
        idx = np.asarray(idx, dtype=np.int64).ravel()
        mine = idx[(idx >= self.lo) & (idx < self.hi)] - self.lo
        if mine.size:
            self.local.drop(mine)

    def reassign(self, dead_ids, normals, offsets, abs_tol=1e-7):
        import torch
        id0, cnt, am, mx = self.local.reassign(dead_ids, normals, offsets, abs_tol)
        if self.world == 1:
            return id0, cnt, np.where(am >= 0, am + self.lo, -1), mx
        big = np.iinfo(np.int64).max
        gidx = np.where(am >= 0, am + self.lo, big)
        pack = torch.as_tensor(np.stack([cnt.astype(np.int64), np.where(am >= 0, mx, -1.0).view(np.int64), gidx],
                                        axis=1))                                    # [n_new, 3]
        backend = self._dist.get_backend()
        if backend != "gloo":
            pack = pack.to(torch.device("cuda", torch.cuda.current_device()))
        allp = torch.empty((self.world * pack.shape[0], 3), dtype=torch.int64, device=pack.device)
        self._dist.all_gather_into_tensor(allp, pack.contiguous())
        allp = allp.cpu().numpy().reshape(self.world, -1, 3)
        count = allp[:, :, 0].sum(axis=0)
        dd = allp[:, :, 1].copy().view(np.float64)
        best = dd.max(axis=0)
        cand = np.where(dd == best[None, :], allp[:, :, 2], big)
        g = cand.min(axis=0)
        none = (g == big) | (best < 0)
        return id0, count, np.where(none, -1, g), np.where(none, 0.0, best)

    def read(self):
        owner, dist_ = self.local.read()
        return owner, dist_

    def close(self):
        self.local.close()


def quickhull_sharded(POINTS, abs_tol=1e-7, session_factory=None):
    """quickhull() with the points' outside sets sharded over the ranks; every rank returns the hull."""
    from . import quickhull as qh
    return qh.quickhull(POINTS, abs_tol=abs_tol,
                        session_factory=lambda X0: ShardedHullSession(X0, session_factory=session_factory))
