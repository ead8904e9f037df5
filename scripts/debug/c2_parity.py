"""C2 batch (100k polytopes, m=16, d=3) through the fused reduce vs the oracle on the first N polytopes: keep / flags / nlp exact."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from oracle import oracle as O
O.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m, d = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 3)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
A, b = random_hpolytopes(B, m, d, seed=0, stream=0)
res = pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda())
torch.cuda.synchronize()
keep = res["keep"].cpu().numpy(); nlp = res["nlp"].cpu().numpy(); fl = res["flags"].cpu().numpy(); r = res["r"].cpu().numpy()
print("nlp mean", nlp.mean(), "kept rows mean", np.mean([bin(int(k)).count("1") for k in keep[:5000]]))
bad = 0
o = O.reduce_batch(A[:N], b[:N]) if hasattr(O, "reduce_batch") else None
if o is not None:
    ok = np.array_equal(o["keep"], keep[:N]) and np.array_equal(o["nlp"], nlp[:N]) and np.array_equal(o["flags"], fl[:N])
    print("oracle batch equal:", ok, "max |r - r_oracle|", np.abs(o["r"] - r[:N]).max())
    if not ok:
        w = np.nonzero((o["keep"] != keep[:N]) | (o["nlp"] != nlp[:N]) | (o["flags"] != fl[:N]))[0]
        print("mismatches", len(w), w[:10], [(hex(int(o["keep"][i])), hex(int(keep[i])), o["nlp"][i], nlp[i]) for i in w[:5]])
else:
    kb = pa.keep_to_bool(keep[:N], m)
    for k in range(N):
        q = O.reduce(A[k], b[k])
        if not (np.array_equal(kb[k], q["keep"]) and int(nlp[k]) == q["nlp"]):
            bad += 1
            if bad < 5: print("mismatch", k, kb[k].astype(int), q["keep"].astype(int), nlp[k], q["nlp"])
    print("mismatches", bad, "of", N)
