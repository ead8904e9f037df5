#!/usr/bin/env python3
"""GPU: the verifier behind the LP / Chebyshev / bounding-box batches (plp_verify.hip) on the soak families -- every answer
against the (certified) oracle, status exact, values within 1e-9 of the extent -- and what it costs (PLP_VERIFY=0 in a
child process for the A/B).  Usage: gpurun -- 'python scripts/debug/verify_smoke.py [seed] [time]'"""
import multiprocessing as mp
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL  # noqa: E402


def _lp_task(args):
    from oracle import oracle as O
    c, G, h, m = args
    out = []
    for k in range(G.shape[0]):
        st, x, f, _ = O.lp_solve(c[k], G[k, :m[k]], h[k, :m[k]])
        out.append((st, f if st == 0 else np.nan, float(np.max(np.abs(x))) if st == 0 else 1.0))
    return out


def timing():
    import torch
    import polytope_amd as pa
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    res = {}
    for (B, m, d) in ((100000, 16, 3), (20000, 32, 6), (20000, 64, 8), (5000, 64, 16)):
        A, b, mr = SL.make(rng, B, m, d, "random")
        At, bt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev)
        c = np.zeros((B, d)); c[:, 0] = 1.0
        ct = torch.as_tensor(c).to(dev)
        for name, fn in (("cheby", lambda: pa.cheby_ball_batch(At, bt)), ("bbox", lambda: pa.bbox_batch(At, bt)),
                         ("lp", lambda: pa.lpsolve_batch(ct, At, bt))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            res["%s(%d,%d)x%d" % (name, m, d, B)] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    print("TIMING verify=%s ms: %s" % (os.environ.get("PLP_VERIFY", "1"), res), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "time":
        timing()
        return 0
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    from oracle import oracle as O
    O.build()
    pool = mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2))
    import torch
    import polytope_amd as pa
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for fam in ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]:
        for (d, m, B) in ((2, 10, 800), (3, 16, 3000), (3, 28, 800), (4, 24, 2500), (6, 32, 1500), (8, 64, 600), (13, 40, 400), (16, 64, 300)):
            A, b, mr = SL.make(rng, B, m, d, fam)
            At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mr).to(dev)
            # bounding boxes + Chebyshev balls
            bb = pa.bbox_batch(At, bt, mt)
            ch = pa.cheby_ball_batch(At, bt, m=mt)
            torch.cuda.synchronize()
            st, lb, ub = bb["status"].cpu().numpy(), bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy()
            cs, cr = ch["status"].cpu().numpy(), ch["r"].cpu().numpy()
            refb = SL.oracle_all(pool, "bbox", A, b, mr, chunk=16)
            nb = nc = nh = 0
            first = None
            for k, (lo, hi, bd, so, ro, xn) in enumerate(refb):
                okc = int(cs[k]) == so and (so != 0 or abs(cr[k] - ro) <= 1e-9 * max(1.0, abs(ro), xn))
                nc += not okc
                if not okc and first is None:
                    first = ("cheby", k, int(cs[k]), so, cr[k], ro)
                if st[k] != 0:
                    nh += 1
                    continue
                okb = bd == 0 and SL.box_equal(lb[k], ub[k], lo, hi)
                nb += not okb
                if not okb and first is None:
                    first = ("bbox", k, lb[k], lo, ub[k], hi)
            # generic LPs: random costs on the same rows
            c = rng.standard_normal((B, d))
            r = pa.lpsolve_batch(torch.as_tensor(c).to(dev), At, bt, mt)
            torch.cuda.synchronize()
            ls, lf = r["status"].cpu().numpy(), r["fun"].cpu().numpy()
            tasks = [(c[i:i + 32], A[i:i + 32], b[i:i + 32], mr[i:i + 32]) for i in range(0, B, 32)]
            ref = [t for part in pool.imap(_lp_task, tasks, chunksize=1) for t in part]
            nl = 0
            for k, (so, fo, ext) in enumerate(ref):
                ok = int(ls[k]) == so and (so != 0 or abs(lf[k] - fo) <= 1e-9 * max(1.0, ext, abs(fo)))
                nl += not ok
                if not ok and first is None:
                    first = ("lp", k, int(ls[k]), so, lf[k], fo)
            bad += nb + nc + nl
            print("%-9s d %2d m %2d B %5d  bbox bad %d (handed %d)  cheby bad %d  lp bad %d  %s" % (
                fam, d, m, B, nb, nh, nc, nl, "" if first is None else first), flush=True)
    print("VERIFY SMOKE %s: %d mismatches, %.0f s" % ("FAILED" if bad else "OK", bad, time.time() - t0), flush=True)
    for v in ("1", "0"):
        subprocess.call([sys.executable, os.path.abspath(__file__), "0", "time"], env=dict(os.environ, PLP_VERIFY=v))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
