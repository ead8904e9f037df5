"""reduce_wsplit_kernel against reduce_wdense_kernel, small batches first (run each under `timeout`):
   python scripts/debug/wsplit_check.py B m d"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth
B, m, d = (int(x) for x in sys.argv[1:4])
A, b = synth.random_hpolytopes(B, m, d, seed=2)
At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
os.environ["PLP_REDUCE_WSPLIT"] = "0"
r0 = {k: v.cpu().numpy() for k, v in pa.reduce_batch(At, bt).items()}
print("wdense done", flush=True)
os.environ["PLP_REDUCE_WSPLIT"] = sys.argv[4] if len(sys.argv) > 4 else "2"
r1 = {k: v.cpu().numpy() for k, v in pa.reduce_batch(At, bt).items()}
print("wsplit done", flush=True)
for k in r0:
    same = np.array_equal(r0[k], r1[k], equal_nan=True) if r0[k].dtype.kind == "f" else np.array_equal(r0[k], r1[k])
    print(k, "equal" if same else "DIFFERENT", "" if same else (r0[k][:4], r1[k][:4]))
