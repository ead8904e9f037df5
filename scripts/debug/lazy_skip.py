import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["PLP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_variants", "libplp_hip_skip.so")
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (B, m, d) in [(20000, 64, 16), (20000, 64, 12), (20000, 48, 9)]:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    pa.reduce_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.reduce_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    print(B, m, d, "F1 + dedupe only: %.3f ms" % (ev[0].elapsed_time(ev[1]) / 5))
