#!/usr/bin/env python3
"""Soak of the stand-alone LP batches (plp_lp_solve_batch: solvers.lpsolve semantics, n = 1..17 columns, up to 64 rows) on the data
families of scripts/soak_lane.py with random costs and feasible sets moved off the origin (phase 1 runs): status exact, objective
1e-9 against the oracle; on the `dup` family scipy / HiGHS -- called as solvers.py:152-154 does -- arbitrates.
Usage: gpurun --timeout 1500 -- 'python scripts/soak_lp.py [trials] [seed]'"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL  # noqa: E402


def _task(args):
    from oracle import oracle as O
    c, G, h, m = args
    out = []
    for k in range(G.shape[0]):
        st, x, fun, _it = O.lp_solve(c[k], G[k, :m[k]], h[k, :m[k]])
        out.append((int(st), float(fun) if st == 0 else float("nan")))
    return out


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from oracle import oracle as O
    O.build()
    pool = mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2))
    import torch
    import polytope_amd as pa
    from scipy.optimize import linprog
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
    bad = nlp = n_off = 0
    t0 = time.time()
    for trial in range(trials):
        d = int(rng.choice([1, 2, 3, 3, 4, 5, 6, 8, 9, 12, 16, 17]))
        m = int(rng.integers(max(2, d), 65))
        B = [int(rng.integers(1, 200)), int(rng.integers(1000, 4000)), int(rng.integers(6000, 20000))][trial % 3]
        if d >= 12 or m > 48:
            B = min(B, 6000)
        fam = fams[int(rng.integers(0, len(fams)))]
        G, h, mrows = SL.make(rng, B, m, d, fam)
        if trial % 2:   # off the origin: the origin is infeasible for many of them, phase 1 runs
            h = h + np.einsum("bij,bj->bi", G, rng.standard_normal((B, d)) * 2.0)
        c = rng.standard_normal((B, d))
        if trial % 5 == 0:
            c[:, rng.integers(0, d)] = 0.0
        res = pa.lpsolve_batch(torch.as_tensor(c).to(dev), torch.as_tensor(G).to(dev), torch.as_tensor(h).to(dev), torch.as_tensor(mrows).to(dev))
        torch.cuda.synchronize()
        st, fun = res["status"].cpu().numpy(), res["fun"].cpu().numpy()
        chunk = 32
        tasks = [(c[i:i + chunk], G[i:i + chunk], h[i:i + chunk], mrows[i:i + chunk]) for i in range(0, B, chunk)]
        ref = [r for part in pool.imap(_task, tasks, chunksize=1) for r in part]
        nb, first = 0, None
        for k, (so, fo) in enumerate(ref):
            ok = int(st[k]) == so and (so != 0 or abs(fun[k] - fo) <= 1e-9 * max(1.0, abs(fo)))
            if not ok and fam == "dup":
                rs = linprog(c[k], G[k, :mrows[k]], h[k, :mrows[k]], None, None, bounds=(None, None))
                ok = rs.status == int(st[k]) and (rs.status != 0 or abs(rs.fun - fun[k]) <= 1e-6 * max(1.0, abs(fun[k])))
                n_off += int(ok)
            if not ok:
                nb += 1
                first = first if first is not None else (k, int(st[k]), so, fun[k], fo)
        nlp += B
        bad += nb
        print("trial %3d  n %2d m %2d B %6d  %-9s moved %d  bad %d   %s" % (trial, d, m, B, fam, trial % 2, nb, "" if first is None else first), flush=True)
    print("LP SOAK %s: %d LPs, %d mismatches, %.0f s  (on nearly duplicated rows HiGHS sides with the kernel against the oracle: %d)" % (
        "FAILED" if bad else "OK", nlp, bad, time.time() - t0, n_off), flush=True)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
