"""Fused reduce at mid shapes: default dispatch vs env-selected variants (times, LP/s, outputs equal)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
VARS = [("default", {}), ("split", {"PLP_REDUCE_SPLIT": "1"}), ("nosplit", {"PLP_REDUCE_SPLIT": "0"}), ("lazy", {"PLP_REDUCE_LAZY": "1"}), ("r2", {"PLP_REDUCE_MIDR2": "1"})]
KEYS = ("PLP_REDUCE_SPLIT", "PLP_REDUCE_LAZY", "PLP_REDUCE_MIDR2")
def run(A, b, env):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    res = pa.reduce_batch(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.reduce_batch(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return res, ev[0].elapsed_time(ev[1]) / 5
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (B, m, d) in shapes:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
    out = []
    r0 = None
    for name, env in VARS:
        r, t = run(A, b, env)
        if r0 is None: r0 = r
        same = all(torch.equal(r0[k], r[k]) for k in ("keep", "flags", "nlp"))
        out.append("%s %.3f ms (%.3g LP/s)%s" % (name, t, int(r["nlp"].sum()) / t * 1e3, "" if same else " DIFFERS"))
    print("B=%d m=%d d=%d: " % (B, m, d) + " | ".join(out), flush=True)
