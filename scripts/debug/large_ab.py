"""The four large-shape configurations of VERDICT r3 item 5, timed with device events, plus a hash of every output array:
two libraries (PLP_LIB) that print the same hashes give the same bits.
  stand-alone F1 (64,16) B = 20 000, two-phase (64,16) B = 20 000, fused reduce (64,8) B = 5 000, (32,6) B = 20 000
(+ (64,12) / (64,16) reduce and (48,9) LPs as by-standers).      python scripts/debug/large_ab.py [reps]"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pa  # noqa: E402
from polytope_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timeit(fn, reps=reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(min(ts))


def digest(res):
    h = hashlib.sha256()
    items = res.items() if isinstance(res, dict) else enumerate(res if isinstance(res, (tuple, list)) else [res])
    for k, v in sorted(items, key=lambda kv: str(kv[0])):
        if torch.is_tensor(v):
            h.update(str(k).encode())
            h.update(np.ascontiguousarray(v.cpu().numpy()).tobytes())
    return h.hexdigest()[:12]


out = {}
for (B, m, d) in [(20000, 64, 16), (20000, 64, 12), (20000, 48, 9), (20000, 64, 8)]:
    A, b = synth.random_hpolytopes(B, m, d, seed=1)
    At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
    res = pa.cheby_ball_batch(At, bt)
    med, mn = timeit(lambda: pa.cheby_ball_batch(At, bt))
    out["F1 (%d,%d) B=%d" % (m, d, B)] = dict(ms=med, ms_min=mn, lp_per_s=B / (med * 1e-3), sha=digest(res))
    rng = np.random.default_rng(3)   # (the set-up of scripts/bench_configs.py lp(): the polytope moved off the origin, phase 1 runs)
    b2t = torch.as_tensor(b + np.einsum("bij,bj->bi", A, rng.standard_normal((B, d)) * 3.0)).to(dev)
    ct = torch.as_tensor(rng.standard_normal((B, d))).to(dev)
    res = pa.lpsolve_batch(ct, At, b2t)
    med, mn = timeit(lambda: pa.lpsolve_batch(ct, At, b2t))
    out["two-phase (%d,%d) B=%d" % (m, d, B)] = dict(ms=med, ms_min=mn, lp_per_s=B / (med * 1e-3), sha=digest(res))
for (B, m, d) in [(5000, 64, 8), (20000, 32, 6), (20000, 32, 8), (5000, 64, 12), (5000, 64, 16), (20000, 64, 8), (1000, 64, 8)]:
    A, b = synth.random_hpolytopes(B, m, d, seed=2)
    At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
    res = pa.reduce_batch(At, bt)
    nlp = int(res["nlp"].sum().item())
    med, mn = timeit(lambda: pa.reduce_batch(At, bt))
    out["reduce (%d,%d) B=%d" % (m, d, B)] = dict(ms=med, ms_min=mn, lp_per_s=nlp / (med * 1e-3), sha=digest(res))
for k, v in out.items():
    print("%-28s %8.4f ms (min %8.4f)  %.4g LP/s  %s" % (k, v["ms"], v["ms_min"], v["lp_per_s"], v["sha"]))
print(json.dumps({"lib": os.environ.get("PLP_LIB", "in-tree"), "results": out}))
