"""GPU: fused reduce at 17..32 rows, d <= 3 (Polytope.intersect of two 16-row polytopes stacks 32): time per batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from polytope_amd import batch
dev = torch.device("cuda:0")
for (m, d) in [(16, 4), (12, 4), (9, 4), (32, 4), (20, 4), (16, 3), (32, 3)]:
    for B in (50000, 10000, 1000):
        A, b = random_hpolytopes(B, m, d, seed=m + d)
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        out = []
        outs = []
        for env in ({"PLP_REDUCE_LANE": "0"}, {}, {"PLP_REDUCE_LANE_GS": "8"}, {"PLP_REDUCE_LANE_GS": "16"}):
            for k in ("PLP_REDUCE_LANE", "PLP_REDUCE_LANE_GS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            batch.reduce_simplex_runs(reset=True)
            res = pa.reduce_batch(At, bt)
            outs.append({k: v.cpu().numpy() for k, v in res.items()})
            ns = batch.reduce_simplex_runs(reset=True)
            for _ in range(3):
                pa.reduce_batch(At, bt)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    pa.reduce_batch(At, bt)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            nlp = int(res["nlp"].sum().item())
            out.append("%s %.1f us (%.2e LP/s, %.1f run)" % (env.get("PLP_REDUCE_LANE_GS", "groups" if env else "default"), best * 1e3, nlp / best * 1e3, ns / B))
        same = all(all(np.array_equal(o[k], outs[0][k]) for k in ("keep", "flags", "nlp")) and np.allclose(o["r"], outs[0]["r"], rtol=0, atol=1e-12, equal_nan=True) for o in outs[1:])
        from oracle import oracle as O
        n = min(B, 3000)
        R = O.reduce_batch(A[:n], b[:n])
        vs = np.array_equal(outs[1]["keep"][:n].view(np.uint64), R["keep"]) and np.array_equal(outs[1]["nlp"][:n], R["nlp"]) and np.array_equal(outs[1]["flags"][:n], R["flags"])
        print("(%d,%d) B=%6d  " % (m, d, B) + "   ".join(out), " same verdicts:", same, " == oracle:", vs, flush=True)
