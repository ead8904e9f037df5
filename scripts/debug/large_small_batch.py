import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (m, d) in [(64, 16), (64, 12), (40, 10), (32, 12), (20, 9)]:
    for B in (1, 16, 256, 2048):
        A, b = random_hpolytopes(B, m, d, seed=2 + B, stream=0, bounded=True)
        At = torch.as_tensor(A).cuda(); bt = torch.as_tensor(b).cuda()
        out = []
        for env in ({}, {"PLP_REDUCE_LAZY": "0"}):
            os.environ.pop("PLP_REDUCE_LAZY", None); os.environ.update(env)
            for _ in range(5): pa.reduce_batch(At, bt)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(30): pa.reduce_batch(At, bt)
            ev[1].record(); torch.cuda.synchronize()
            out.append(ev[0].elapsed_time(ev[1]) / 30 * 1e3)
        print("(%d,%d) B=%-5d default %.0f us, two rows per lane %.0f us" % (m, d, B, out[0], out[1]), flush=True)
