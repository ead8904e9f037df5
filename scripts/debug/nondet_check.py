import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
A, b = random_hpolytopes(B, 16, 3, seed=0, stream=0)
At=torch.as_tensor(A).cuda(); bt=torch.as_tensor(b).cuda()
ref=None
for it in range(6):
    res = pa.reduce_batch(At, bt)
    torch.cuda.synchronize()
    cur={k:res[k].cpu().numpy().copy() for k in ("keep","nlp","flags","r","xc")}
    if ref is None: ref=cur; continue
    bad=np.nonzero((cur["keep"]!=ref["keep"])|(cur["nlp"]!=ref["nlp"])|np.any(cur["xc"]!=ref["xc"],axis=1))[0]
    print("iter",it,"differs from iter 0 at",len(bad),"polytopes; tiles",sorted(set(bad//16))[:10])
    for i in bad[:3]:
        print("  ",i,hex(int(ref["keep"][i])),hex(int(cur["keep"][i])),ref["nlp"][i],cur["nlp"][i],ref["flags"][i],cur["flags"][i],ref["r"][i],cur["r"][i],ref["xc"][i],cur["xc"][i])
