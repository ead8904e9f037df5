"""reduce_wsplit_kernel (NW wavefronts per polytope) against reduce_wdense_kernel over batch sizes: ms and equality of outputs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa
from polytope_amd import synth

def timeit(fn, reps=9):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for (m, d) in [(64, 8), (32, 6), (64, 12)]:
    for B in [1, 250, 1000, 1500, 2000, 3000, 5000, 8000, 12000, 16000]:
        A, b = synth.random_hpolytopes(B, m, d, seed=2)
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        out = {}
        for v in ("0", "2", "4"):
            os.environ["PLP_REDUCE_WSPLIT"] = v
            res = pa.reduce_batch(At, bt)
            out[v] = ({k: x.cpu().numpy() for k, x in res.items()}, timeit(lambda: pa.reduce_batch(At, bt)))
        same = all(np.array_equal(out["0"][0][k], out[v][0][k], equal_nan=(out["0"][0][k].dtype.kind == "f")) for k in out["0"][0] for v in ("2", "4"))
        print("(%d,%d) B=%-5d wdense %.4f ms  2 waves %.4f ms (x%.2f)  4 waves %.4f ms (x%.2f)  %s" % (m, d, B, out["0"][1], out["2"][1], out["0"][1] / out["2"][1], out["4"][1], out["0"][1] / out["4"][1], "equal" if same else "DIFFERENT"), flush=True)
