import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import polytope_amd as pa
from polytope_amd import synth
from oracle import oracle as O
O.build()
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
for (B, m, d) in [(20000, 64, 16), (20000, 64, 12), (20000, 48, 9), (20000, 64, 8), (20000, 33, 5)]:
    A, b = synth.random_hpolytopes(B, m, d, seed=1)
    # a few unbounded / empty / ragged ones
    b[5] = -1.0
    A[7, :, 0] = np.abs(A[7, :, 0])
    ms = np.full(B, m, np.int32); ms[::3] = m - 2
    At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(ms).to(dev)
    os.environ["PLP_CHEBY_WIDE"] = "0"
    ref = pa.cheby_ball_batch(At, bt, m=mt); t0 = timeit(lambda: pa.cheby_ball_batch(At, bt, m=mt))
    os.environ["PLP_CHEBY_WIDE"] = "1"
    got = pa.cheby_ball_batch(At, bt, m=mt); t1 = timeit(lambda: pa.cheby_ball_batch(At, bt, m=mt))
    rs, gs = ref["status"].cpu().numpy(), got["status"].cpu().numpy()
    rr, gr = ref["r"].cpu().numpy(), got["r"].cpu().numpy()
    ok = rs == 0
    print((B, m, d), "status equal", np.array_equal(rs, gs), "statuses", np.unique(gs), "max |dr| %.2e" % np.nanmax(np.abs(rr[ok] - gr[ok])),
          "lane-group %.3f ms (%.3g LP/s)  wide %.3f ms (%.3g LP/s)" % (t0 * 1e3, B / t0, t1 * 1e3, B / t1))
    xc = got["xc"].cpu().numpy()
    worst = 0
    for k in range(0, 400, 7):
        so, ro, _ = O.cheby(A[k, :ms[k]], b[k, :ms[k]])
        assert so == gs[k], (k, so, gs[k])
        if so == 0:
            worst = max(worst, abs(ro - gr[k]))
            assert np.max(A[k, :ms[k]] @ xc[k] + gr[k] - b[k, :ms[k]]) < 1e-8
    print("   oracle: worst |dr| %.2e" % worst)
