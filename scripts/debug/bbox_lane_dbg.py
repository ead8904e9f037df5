import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from oracle import oracle as O
dev = torch.device("cuda:0")
m, d, B = 24, 2, 100000
A, b = random_hpolytopes(B, m, d, seed=m + d)
b = b + np.einsum("bij,bj->bi", A, np.random.default_rng(1).standard_normal((B, d)))
At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
res = []
for lane in ("0", "1"):
    os.environ["PLP_BBOX_LANE"] = lane
    r = pa.bbox_batch(At, bt); res.append({k: v.cpu().numpy() for k, v in r.items()})
g, l = res
ds = np.nonzero(g["status"] != l["status"])[0]
print("status differs at", ds[:10], "groups", g["status"][ds[:10]], "lane", l["status"][ds[:10]])
dv = np.abs(np.where(np.isfinite(g["lb"]) & np.isfinite(l["lb"]), g["lb"] - l["lb"], 0)).max(axis=1) + np.abs(np.where(np.isfinite(g["ub"]) & np.isfinite(l["ub"]), g["ub"] - l["ub"], 0)).max(axis=1)
w = np.argsort(-dv)[:4]
for k in list(w) + list(ds[:3]):
    lb, ub, bad = O.bounding_box(A[k], b[k])
    print(k, "status g/l", g["status"][k], l["status"][k], "oracle bad", bad)
    print("  oracle lb", lb, "ub", ub)
    print("  groups lb", g["lb"][k], "ub", g["ub"][k])
    print("  lane   lb", l["lb"][k], "ub", l["ub"][k])
