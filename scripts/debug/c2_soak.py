"""Bench kernel under load: several C2-size batches (different seeds), each reduced 8 times back to back on two streams
(4 waves per SIMD resident, launches overlapping) -- every repetition bit-identical to the first, and the first equal
to the oracle on a sample.  Guards against the spill-related nondeterminism seen in an instrumented build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
from oracle import oracle as O
O.build()
bad = 0
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    A, b = random_hpolytopes(100000, 16, 3, seed=seed, stream=seed % 3)
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    ref = pa.reduce_batch(At, bt); torch.cuda.synchronize()
    outs = []
    for rep in range(8):
        with torch.cuda.stream(streams[rep & 1]):
            outs.append(pa.reduce_batch(At, bt))
    torch.cuda.synchronize()
    for r in outs:
        for k in ("keep", "flags", "nlp"):
            bad += int(not torch.equal(r[k], ref[k]))
        bad += int(not torch.equal(r["r"].view(torch.int64), ref["r"].view(torch.int64)))
    keep = pa.keep_to_bool(ref["keep"].cpu().numpy(), 16); nlp = ref["nlp"].cpu().numpy()
    for k in range(seed * 1000, seed * 1000 + 4000):
        q = O.reduce(A[k], b[k])
        bad += int(not (np.array_equal(keep[k], q["keep"]) and int(nlp[k]) == q["nlp"]))
    print("seed", seed, "bad so far", bad, flush=True)
print("C2 SOAK", "FAILED" if bad else "OK")
