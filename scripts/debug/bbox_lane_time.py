"""GPU: plp_bbox_batch at (<= 32 rows, d <= 3): bbox_lane_kernel (one LP per lane) against the lane-group kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
dev = torch.device("cuda:0")
for (m, d) in [(16, 3), (32, 3), (12, 2), (24, 2)]:
    for B in (100000, 20000, 2000, 64):
        A, b = random_hpolytopes(B, m, d, seed=m + d)
        b = b + np.einsum("bij,bj->bi", A, np.random.default_rng(1).standard_normal((B, d)))   # boxes away from the origin
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        out, res = [], []
        for lane in ("0", "1"):
            os.environ["PLP_BBOX_LANE"] = lane
            r = pa.bbox_batch(At, bt)
            res.append({k: v.cpu().numpy() for k, v in r.items()})
            for _ in range(3):
                pa.bbox_batch(At, bt)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    pa.bbox_batch(At, bt)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            out.append("%s %.1f us" % ("lane" if lane == "1" else "groups", best * 1e3))
        os.environ.pop("PLP_BBOX_LANE")
        keys = list(res[0].keys())
        same_status = all(np.array_equal(res[0][k], res[1][k]) for k in keys if "status" in k)
        err = max(float(np.nanmax(np.abs(np.where(np.isfinite(res[0][k]), res[0][k] - res[1][k], 0.0)))) for k in keys if "status" not in k)
        print("(%d,%d) B=%6d  %s   status equal: %s  max |lb/ub diff| %.1e  keys %s" % (m, d, B, "   ".join(out), same_status, err, keys), flush=True)
