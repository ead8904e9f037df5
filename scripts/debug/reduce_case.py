"""One polytope (npz with A, b) through the fused reduce at several batch sizes / engine choices, next to the oracle.
Usage: gpurun -- 'python scripts/debug/reduce_case.py case.npz'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import polytope_amd as pa  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
z = np.load(sys.argv[1])
A, b = z["A"], z["b"]
m, d = A.shape
ref = O.reduce(A, b)
print("oracle  keep %s flags %d nlp %d r %r" % (hex(ref["mask"]), ref["flags"], ref["nlp"], ref["r"]))
dev = torch.device("cuda:0")
for env in ({}, {"PLP_REDUCE_LANE": "1"}, {"PLP_REDUCE_LANE": "0"}):
    for k_ in ("PLP_REDUCE_LANE", "PLP_REDUCE_LANE_GS"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    for B in (1, 100, 36000):
        At = torch.as_tensor(np.repeat(A[None], B, 0)).to(dev)
        bt = torch.as_tensor(np.repeat(b[None], B, 0)).to(dev)
        mt = torch.full((B,), m, dtype=torch.int32, device=dev)
        rd = pa.reduce_batch(At, bt, mt)
        torch.cuda.synchronize()
        keep = rd["keep"].cpu().numpy().view(np.uint64)
        print(env, "B", B, "keep", sorted(set(hex(int(k)) for k in keep)), "flags", set(rd["flags"].cpu().numpy().tolist()),
              "nlp", set(rd["nlp"].cpu().numpy().tolist()), "r", set(rd["r"].cpu().numpy().tolist()))
