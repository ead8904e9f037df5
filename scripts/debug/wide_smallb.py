import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import polytope_amd as pa
from polytope_amd import synth
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for (m, d) in [(32, 6), (16, 5), (24, 8), (32, 10)]:
    row = []
    for B in (3000, 4096, 5120, 6144, 8192, 10240, 12288, 16384):
        A, b = synth.random_hpolytopes(B, m, d, seed=1)
        At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
        os.environ["PLP_CHEBY_WIDE"] = "0"; t0 = timeit(lambda: pa.cheby_ball_batch(At, bt))
        os.environ["PLP_CHEBY_WIDE"] = "1"; t1 = timeit(lambda: pa.cheby_ball_batch(At, bt))
        row.append("%d: %.1f/%.1f%s" % (B, t0 * 1e6, t1 * 1e6, "*" if t1 < t0 else ""))
    print((m, d), "  ".join(row), flush=True)
