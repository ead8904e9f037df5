import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from oracle import oracle as O
dev = torch.device("cuda:0")
for (B, m, d) in [(9000, 16, 3), (5000, 11, 2)]:
    A, b = random_hpolytopes(B, m, d, seed=3)
    os.environ["PLP_REDUCE_LANE"] = "1"
    os.environ["PLP_REDUCE_RETRY_ALL"] = "1"
    r = pa.reduce_batch(torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)); torch.cuda.synchronize()
    del os.environ["PLP_REDUCE_RETRY_ALL"]
    R = O.reduce_batch(A, b)
    ok = np.array_equal(r["keep"].cpu().numpy().view(np.uint64), R["keep"]) and np.array_equal(r["flags"].cpu().numpy(), R["flags"]) and np.array_equal(r["nlp"].cpu().numpy(), R["nlp"]) and float(np.abs(r["r"].cpu().numpy() - R["r"]).max()) <= 1e-9
    print("forced hand-over inside the kernel", (B, m, d), "== oracle:", ok)
