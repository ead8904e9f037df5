#!/usr/bin/env python3
"""Soak of the fused reduce BEYOND the lane kernels' shapes (d = 4..16, up to 64 rows: the lane-group kernels with four / two
rows per lane, the one-polytope-per-wavefront kernels, their latency forms) on the data families of scripts/soak_lane.py --
random, ragged, unbounded-allowed, duplicated / nearly duplicated, rescaled, flat, lattice -- EVERY polytope against the
oracle (keep mask, flags, LP count exact; radius 1e-9), the oracle on all host cores.
Usage: gpurun --timeout 1500 -- 'python scripts/soak_wide.py [trials] [seed]'"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from oracle import oracle as O
    O.build()
    pool = mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2))
    import torch
    import polytope_amd as pa
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
    bad = npoly = n_off = n_tie = n_cond = n_open = 0
    t0 = time.time()
    for trial in range(trials):
        d = int(rng.choice([4, 5, 5, 6, 6, 7, 8, 8, 9, 10, 12, 13, 14, 16]))
        m = int(rng.integers(d + 1, 65))
        cls = trial % 5
        B = [int(rng.integers(1, 200)), int(rng.integers(1000, 3000)), int(rng.integers(4000, 9000)),
             int(rng.integers(12000, 22000)), int(rng.integers(300, 1000))][cls]
        if d >= 12 or m > 48:
            B = min(B, 6000)      # (the oracle's share of the time)
        fam = fams[int(rng.integers(0, len(fams)))]
        A, b, mrows = SL.make(rng, B, m, d, fam)
        At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mrows).to(dev)
        rd = pa.reduce_batch(At, bt, mt)
        torch.cuda.synchronize()
        keep = rd["keep"].cpu().numpy().view(np.uint64)
        flags, nlp, r = rd["flags"].cpu().numpy(), rd["nlp"].cpu().numpy(), rd["r"].cpu().numpy()
        ref = SL.oracle_all(pool, "reduce", A, b, mrows, chunk=16)
        nb, first = 0, None
        for k, (mk, fl, nl, rr) in enumerate(ref):
            ok = int(keep[k]) == mk and int(flags[k]) == fl and int(nlp[k]) == nl and \
                (abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr)) or (not np.isfinite(rr) and not np.isfinite(r[k])))
            if not ok and fam == "dup" and int(keep[k]) == mk and int(flags[k]) == fl and int(nlp[k]) == nl:
                ok = SL.highs_radius_agrees(A[k, :mrows[k]], b[k, :mrows[k]], r[k])   # (see soak_lane.py: the oracle's own limit)
                n_off += int(ok)
            if not ok and int(keep[k]) == mk and int(flags[k]) == fl and abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr)) \
                    and SL.prefilter_tie(A[k, :mrows[k]], b[k, :mrows[k]]):
                ok = True          # only the LP count differs, and a row sits on the prefilter's threshold (soak_lane.prefilter_tie)
                n_tie += 1
            if not ok and (int(flags[k]) & 32) and not (fl & 32) and SL.public_reduce_agrees(A[k, :mrows[k]], b[k, :mrows[k]], mk):
                ok = True          # the kernel handed the polytope back (RF_F1OPEN); the public reduce() keeps the oracle's rows
                n_open += 1
            if not ok:
                nb += 1
                first = first if first is not None else (k, hex(int(keep[k])), hex(mk), int(flags[k]), fl, int(nlp[k]), nl, r[k], rr)
        # stand-alone Chebyshev balls (every polytope) and bounding boxes (a part of the batch), HiGHS arbitrating on `dup`
        nc = nbb = 0
        ch = pa.cheby_ball_batch(At, bt, m=mt)
        torch.cuda.synchronize()
        cs, cr = ch["status"].cpu().numpy(), ch["r"].cpu().numpy()
        cxn = np.nan_to_num(np.abs(ch["xc"].cpu().numpy()).max(axis=1), nan=1.0)   # how far the ball's centre lies
        nq = min(B, 1500)
        bb = pa.bbox_batch(At[:nq], bt[:nq], mt[:nq])
        refb = SL.oracle_all(pool, "bbox", A[:nq], b[:nq], mrows[:nq], chunk=16)
        if bb is not None:
            torch.cuda.synchronize()
            st, lb, ub = bb["status"].cpu().numpy(), bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy()
        for k, (lo, hi, bd, so, ro, xn) in enumerate(refb):
            okc = int(cs[k]) == so and (so != 0 or abs(cr[k] - ro) <= 1e-9 * max(1.0, abs(ro), xn, cxn[k]))
            if not okc:
                nc += 1
                first = first if first is not None else ("cheby", k, int(cs[k]), so, cr[k], ro)
            if bb is None or st[k] != 0:
                continue
            okb = bd == 0 and SL.box_equal(lb[k], ub[k], lo, hi, hair_unbounded_tol=(1e-8 if fam == "dup" else None))
            if not okb and bd == 0 and fam == "dup" and SL.box_equal(lb[k], ub[k], lo, hi, tol=SL.hair_tol(A[k, :mrows[k]]), hair_unbounded_tol=1e-8):
                okb = True           # within what two rows a hair apart define (soak_lane.hair_tol): classified, counted
                n_cond += 1
            if not okb:
                nbb += 1
                first = first if first is not None else ("bbox", k, lb[k], lo, ub[k], hi, bd)
        npoly += B
        bad += nb + nc + nbb
        print("trial %3d  d %2d m %2d B %6d  %-9s reduce bad %d  cheby bad %d  bbox bad %d   %s" % (
            trial, d, m, B, fam, nb, nc, nbb, "" if first is None else first), flush=True)
    print("WIDE SOAK %s: %d polytopes, %d mismatches, %.0f s  (radii on nearly duplicated rows where HiGHS sides with the fused kernel "
          "against the oracle's raw engine: %d; LP counts that differ on a prefilter tie: %d; boxes within the conditioning of rows a hair "
          "apart: %d; handed back by the fused kernel (RF_F1OPEN) and right through the public reduce(): %d)" % (
              "FAILED" if bad else "OK", npoly, bad, time.time() - t0, n_off, n_tie, n_cond, n_open), flush=True)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
