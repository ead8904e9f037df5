"""The mixed-tile launch of the fused reduce (PLP_REDUCE_MIX / PLP_REDUCE_MIXQ) against full tiles only: every output
array bitwise equal on a C2 batch.  Env is read per launch, so one process can switch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
A, b = random_hpolytopes(100000, 16, 3, seed=3, stream=0)
A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
def run(mix, mixq):
    os.environ["PLP_REDUCE_MIX"] = str(mix)
    if mixq is None: os.environ.pop("PLP_REDUCE_MIXQ", None)
    else: os.environ["PLP_REDUCE_MIXQ"] = str(mixq)
    r = pa.reduce_batch(A, b)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in r.items()}
ref = run(0, None)
for mix, mixq in ((4, None), (4, 4), (4, 16), (8, 32), (2, 8)):
    got = run(mix, mixq)
    ok = all(torch.equal(ref[k].view(torch.uint8), got[k].view(torch.uint8)) for k in ref)
    print("MIX=%s MIXQ=%s bitwise equal: %s" % (mix, mixq, ok))
    assert ok
