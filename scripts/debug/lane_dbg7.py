import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from oracle import oracle as O
rng = np.random.default_rng(17)
m,d,B=7,3,3000
A, b = random_hpolytopes(B, m, d, seed=13 * m + d + B, bounded=True)
for k in range(0, B, 5):
    j = int(rng.integers(m))
    A[k, (j + 1) % m] = A[k, j]
    b[k, (j + 1) % m] = b[k, j] + rng.choice([0.0, 0.05])
for k in range(3, B, 11):
    b[k, 0] = -4.0
def run(**env):
    for k in ("PLP_REDUCE_LANE","PLP_REDUCE_SPLIT","PLP_REDUCE_HALF"): os.environ.pop(k,None)
    os.environ.update(env)
    r=pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()); torch.cuda.synchronize()
    return {k:v.cpu().numpy() for k,v in r.items()}
l=run(); g=run(PLP_REDUCE_LANE="0"); g2=run(PLP_REDUCE_LANE="0",PLP_REDUCE_SPLIT="0",PLP_REDUCE_HALF="0")
for k in l:
    d1=np.nonzero((l[k].view(np.int64 if l[k].dtype.itemsize==8 else np.int32)!=g[k].view(np.int64 if l[k].dtype.itemsize==8 else np.int32)).reshape(B,-1).any(axis=1))[0]; d2=np.nonzero((l[k].view(np.int64 if l[k].dtype.itemsize==8 else np.int32)!=g2[k].view(np.int64 if l[k].dtype.itemsize==8 else np.int32)).reshape(B,-1).any(axis=1))[0]
    print(k, "lane vs groups(default)", len(d1), d1[:5], " vs groups(batch form)", len(d2))
k=d1[0] if len(d1) else 0
o=O.reduce(A[k],b[k]); print(o["r"], o["xc"], o["flags"], o["mask"]); print(A[k], b[k]); print(l["xc"][k], g["xc"][k], l["r"][k], g["r"][k], l["flags"][k], g["flags"][k], l["keep"][k], g["keep"][k])
dr=np.abs(l["r"]-g["r"]); print("max |dr|", np.nanmax(dr))
dx=np.abs(l["xc"]-g["xc"]); print("max |dxc|", np.nanmax(dx), np.argwhere(dx>1e-9)[:5])
for gs in ("4","8","16"):
    x=run(PLP_REDUCE_LANE="1",PLP_REDUCE_LANE_GS=gs)
    print("GS",gs,"r diff vs groups:", int((x["r"].view(np.int64)!=g["r"].view(np.int64)).sum()), "xc:", int((x["xc"].view(np.int64)!=g["xc"].view(np.int64)).any(axis=1).sum()))
os.environ.pop("PLP_REDUCE_LANE_GS",None)
