"""Regenerate the batch of one trial of scripts/soak_wide.py (same RNG stream); with `gpu`: the fused reduce and the Chebyshev
ball of the listed polytopes on the device against the oracle and HiGHS.   python scripts/debug/soak_wide_repro.py <seed> <trial> [gpu k...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
for trial in range(want + 1):
    d = int(rng.choice([4, 5, 5, 6, 6, 7, 8, 8, 9, 10, 12, 13, 14, 16]))
    m = int(rng.integers(d + 1, 65))
    cls = trial % 5
    B = [int(rng.integers(1, 200)), int(rng.integers(1000, 3000)), int(rng.integers(4000, 9000)),
         int(rng.integers(12000, 22000)), int(rng.integers(300, 1000))][cls]
    if d >= 12 or m > 48:
        B = min(B, 6000)
    fam = fams[int(rng.integers(0, len(fams)))]
    A, b, mrows = SL.make(rng, B, m, d, fam)
print("trial", want, "d", d, "m", m, "B", B, fam)
ks = [int(x) for x in sys.argv[4:]] if len(sys.argv) > 4 else []
from oracle import oracle as O
O.build()
from scipy.optimize import linprog
np.set_printoptions(linewidth=200, precision=12)
for k in ks:
    Ak, bk = A[k, :mrows[k]], b[k, :mrows[k]]
    st, r, xc = O.cheby(Ak, bk)
    nrm = np.sqrt((Ak * Ak).sum(1))
    c = np.zeros(d + 1); c[d] = -1
    rs = linprog(c, np.hstack([Ak, nrm[:, None]]), bk, bounds=(None, None))
    print("poly", k, "oracle cheby st", st, "r", r, " HiGHS", rs.status, -rs.fun if rs.status == 0 else None)
    An = Ak / nrm[:, None]; G = An @ An.T
    for i in range(len(bk)):
        for j in range(i + 1, len(bk)):
            if G[i, j] > 1 - 1e-6: print("   near-dup rows", i, j, "1-dot %.2e" % (1 - G[i, j]), "b/n", bk[i] / nrm[i], bk[j] / nrm[j])
if len(sys.argv) > 3 and sys.argv[3] == "gpu":
    import torch, polytope_amd as pa
    dev = torch.device("cuda:0")
    At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mrows).to(dev)
    rd = pa.reduce_batch(At, bt, mt)
    ch = pa.cheby_ball_batch(At, bt, m=mt)
    for k in ks:
        print("poly", k, "reduce: keep", hex(int(rd["keep"][k].cpu().numpy().view(np.uint64))), "flags", int(rd["flags"][k]), "nlp", int(rd["nlp"][k]), "r", float(rd["r"][k]),
              "| cheby_ball_batch: status", int(ch["status"][k]), "r", float(ch["r"][k]))
        one = pa.cheby_ball_batch(At[k:k + 1], bt[k:k + 1], m=mt[k:k + 1])
        print("     alone: status", int(one["status"][0]), "r", float(one["r"][0]))
        for env in ({"PLP_CHEBY_WIDE": "0"}, {"PLP_CHEBY_WIDE": "1"}):
            os.environ.update(env)
            one = pa.cheby_ball_batch(At[k:k + 1], bt[k:k + 1], m=mt[k:k + 1])
            print("     ", env, "status", int(one["status"][0]), "r", float(one["r"][0]))
            for e in env: os.environ.pop(e)
if len(sys.argv) > 3 and sys.argv[3] == "bbox":
    import torch, polytope_amd as pa
    from scipy.optimize import linprog
    dev = torch.device("cuda:0")
    ks = [int(x) for x in sys.argv[4:]]
    for k in ks:
        Ak, bk = A[k:k + 1, :mrows[k]], b[k:k + 1, :mrows[k]]
        lo, hi, bd = O.bounding_box(Ak[0], bk[0])
        hl, hh = [], []
        for i in range(d):
            for sgn, dst in ((1.0, hl), (-1.0, hh)):
                c = np.zeros(d); c[i] = sgn
                rs = linprog(c, Ak[0], bk[0], bounds=(None, None))
                dst.append(rs.x[i] if rs.status == 0 else (-np.inf if sgn > 0 else np.inf))
        print("poly", k, "\n  oracle lb", lo, "\n  HiGHS  lb", np.array(hl), "\n  oracle ub", hi, "\n  HiGHS  ub", np.array(hh))
        for env in ({}, {"PLP_BBOX_SPLIT": "0"}, {"PLP_BBOX_SPLIT": "1"}):
            os.environ.update(env)
            for B in (1, 3000):
                bb = pa.bbox_batch(torch.as_tensor(np.repeat(Ak, B, 0)).to(dev), torch.as_tensor(np.repeat(bk, B, 0)).to(dev))
                print("  kernel", env, "B", B, "status", int(bb["status"][0]), "lb", bb["lb"][0].cpu().numpy(), "ub", bb["ub"][0].cpu().numpy())
            for e in env: os.environ.pop(e)
        np.savez("gpurun_out/bbox_repro_%d_%d_%d.npz" % (seed, want, k), A=Ak[0], b=bk[0])
