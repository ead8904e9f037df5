"""contains_batch at several d (1M points x 2000 polytopes of 16 rows): ms per call (A/B of points per lane via PLP_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd import synth
out = []
for d in (tuple(int(v) for v in os.environ["PPL_DIMS"].split(",")) if os.environ.get("PPL_DIMS") else (2, 3, 4, 5, 6, 7, 8)):
    A, b, X = synth.containment_workload(2000, 1000000, d=d, m=max(16, 2 * d + 8), seed=0)
    At, bt, Xt = (torch.as_tensor(v).cuda() for v in (A, b, X))
    pa.contains_batch(At, bt, Xt, 1e-7); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3): pa.contains_batch(At, bt, Xt, 1e-7)
    ev[1].record(); torch.cuda.synchronize()
    out.append("d=%d %.2f" % (d, ev[0].elapsed_time(ev[1]) / 3))
print(os.path.basename(os.environ.get("PLP_LIB", "default")), " ".join(out), flush=True)
