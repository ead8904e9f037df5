"""Medium batches of the fused reduce at (16,3): full tiles, half-size tiles only (PLP_REDUCE_HALF=1), latency form."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
for (m, d) in [(16, 3), (12, 4)]:
    for B in (4097, 6000, 8192, 12500, 16384, 25000, 32768, 50000, 65536):
        A, b = random_hpolytopes(B, m, d, seed=2, stream=0)
        At = torch.as_tensor(A).cuda(); bt = torch.as_tensor(b).cuda()
        res = {}
        for name, env in (("full", {"PLP_REDUCE_HALF": "0", "PLP_REDUCE_SPLIT": "0"}), ("half", {"PLP_REDUCE_HALF": "1", "PLP_REDUCE_SPLIT": "0"}),
                          ("latency", {"PLP_REDUCE_SPLIT": "1"})):
            for k in ("PLP_REDUCE_HALF", "PLP_REDUCE_SPLIT"): os.environ.pop(k, None)
            os.environ.update(env)
            out = pa.reduce_batch(At, bt); torch.cuda.synchronize()
            for _ in range(5): pa.reduce_batch(At, bt)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(50): pa.reduce_batch(At, bt)
            ev[1].record(); torch.cuda.synchronize()
            res[name] = (out, ev[0].elapsed_time(ev[1]) * 20)
        same = all(torch.equal(res["full"][0][k].view(torch.uint8), res["half"][0][k].view(torch.uint8)) for k in res["full"][0])
        print("(%d,%d) B=%-6d full tiles %.1f us, half tiles %.1f us, latency form %.1f us, half == full %s" % (m, d, B, res["full"][1], res["half"][1], res["latency"][1], same), flush=True)
