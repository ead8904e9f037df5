import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from polytope_amd.dist import ResultBuffer
dev = torch.device("cuda:0")
A, b = random_hpolytopes(100000, 16, 3, seed=0)
Ap, bp = torch.as_tensor(A).pin_memory(), torch.as_tensor(b).pin_memory()
Ad, bd = torch.empty_like(Ap, device=dev), torch.empty_like(bp, device=dev)
rb = ResultBuffer(torch, 100000, 3, dev)
host = torch.empty((rb.nbytes,), dtype=torch.uint8).pin_memory()
print("pinned:", host.is_pinned(), Ap.is_pinned())
for i in range(4):
    t = [time.perf_counter()]
    Ad.copy_(Ap, non_blocking=True); bd.copy_(bp, non_blocking=True); t.append(time.perf_counter())
    pa.reduce_batch(Ad, bd, out=rb.views); t.append(time.perf_counter())
    host.copy_(rb.flat, non_blocking=True); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    print(["%.3f" % ((t[k + 1] - t[k]) * 1e3) for k in range(4)])
import bench
for _ in range(3):
    o = bench.end_to_end(torch, pa, A, b, dev)
    print(o["pageable"]["ms_per_pass"], o["pinned"])
