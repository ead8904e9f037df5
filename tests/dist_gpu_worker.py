"""One rank of the multi-rank GPU check (launched by tests/test_dist_gpu.py through torch.distributed.run, gloo
rendezvous, every rank on cuda:0): the sharded paths of polytope_amd.dist with the REAL HIP kernels on every rank --
rank > 0 included -- against the unsharded calls on the same GPU.  Prints 'RANK r OK' or raises."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import polytope_amd as pa  # noqa: E402
from polytope_amd import dist as pdist  # noqa: E402
from polytope_amd import solvers  # noqa: E402
from polytope_amd.synth import containment_workload, random_hpolytopes  # noqa: E402


def _h(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)   # all ranks share the one GPU of the box
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solvers.default_solver = "hip"
    # ---- reduce: contiguous shards of one batch, one fused all-gather of the packed results
    B = 3001   # uneven shards
    A, b = random_hpolytopes(B, 16, 3, seed=11)
    got = pdist.reduce_batch_sharded(A, b, device=dev)
    ref = pa.reduce_batch(torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev))
    for k in ("keep", "flags", "nlp", "r"):
        assert torch.equal(got[k].cpu(), ref[k].cpu()), (rank, k)
    # ---- containment: points sharded, polytopes replicated
    Ac, bc, X = containment_workload(40, 20001, d=6, m=16, seed=3)
    got = pdist.contains_sharded(Ac, bc, X, device=dev)
    ref = pa.contains_batch(torch.as_tensor(Ac).to(dev), torch.as_tensor(bc).to(dev), torch.as_tensor(X).to(dev))
    assert np.array_equal(_h(got), _h(ref)), rank
    # ---- adjacency: the pair index space sharded, one byte per pair gathered
    cells = []
    for i in range(5):
        for j in range(5):
            for k in range(3):
                lo = np.array([i, j, k], float) / 5.0
                cells.append((np.vstack([np.eye(3), -np.eye(3)]), np.r_[lo + 0.2, -lo]))
    Aa = np.stack([c[0] for c in cells]); ba = np.stack([c[1] for c in cells])
    got = pdist.adjacent_pairs_sharded(Aa, ba, device=dev)
    ref = pa.adjacent_pairs(torch.as_tensor(Aa).to(dev), torch.as_tensor(ba).to(dev))
    assert np.array_equal(_h(got), _h(ref)), rank
    # ---- quickhull: points sharded, every rank runs the same facet graph; rows identical to the single-process hull
    P = np.random.default_rng(5).standard_normal((60000, 3))
    np.random.seed(4)
    A1, b1, V1 = pdist.quickhull_sharded(P)
    np.random.seed(4)
    import polytope_amd.quickhull as Q
    from polytope_amd.batch import HullSession
    A0, b0, V0 = Q.quickhull(P, session_factory=lambda X0: HullSession(X0))   # the same host facet graph, one session
    assert np.array_equal(A1, A0) and np.array_equal(b1, b0) and np.array_equal(V1, V0), rank
    np.random.seed(4)
    A2, b2, V2 = Q.quickhull(P)   # the library's main loop (own LU from 4096 points on): same rows to rounding
    assert A2.shape == A0.shape and np.allclose(A2, A0, rtol=0, atol=1e-12) and np.allclose(b2, b0, rtol=0, atol=1e-12), rank
    # ---- furthest-point pass (C5): points sharded, per-facet (max dist, lowest global index) combined
    Xp = np.random.default_rng(6).random((50001, 8))
    nrm = np.random.default_rng(7).standard_normal((9, 8)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    off = nrm @ np.full(8, 0.5) + 0.1
    got = pdist.assign_sharded(Xp, nrm, off, device=dev)
    ref = pa.assign_batch(torch.as_tensor(Xp).to(dev), torch.as_tensor(nrm).to(dev), torch.as_tensor(off).to(dev))
    assert np.array_equal(_h(got["argmax"]), _h(ref["argmax"])), rank
    assert np.array_equal(_h(got["facet"]), _h(ref["facet"])), rank
    dist.barrier()
    print("RANK %d OK" % rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
