import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
B, m, d = 5000, 64, 16
A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
for env in ({}, {"PLP_REDUCE_LAZY": "1"}):
    for k in ("PLP_REDUCE_R1", "PLP_REDUCE_LAZY"): os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3): pa.reduce_batch(A, b)
    torch.cuda.synchronize()
