#!/usr/bin/env python3
"""Soak of the kernels for polytopes of MORE than 64 rows (rows and dictionary in LDS: plp_lds.hip -- reduce_lds_kernel, the LDS
forms of the Chebyshev / LP / bounding-box engines) on the data families of scripts/soak_lane.py: every polytope's fused reduce
(keep words, flags, LP count exact; radius 1e-9), stand-alone ball and box against the certified oracle.
Usage: gpurun --timeout 1500 -- 'python scripts/soak_tall.py [trials] [seed]'"""
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
    A, b, m = args
    out = []
    for k in range(A.shape[0]):
        Ak, bk = A[k, :m[k]], b[k, :m[k]]
        o = O.reduce(Ak, bk)
        lo, hi, bad = O.bounding_box(Ak, bk)[:3]
        so, ro, _ = O.cheby(Ak, bk)
        out.append((tuple(int(w) for w in o["words"]), int(o["flags"]), int(o["nlp"]), float(o["r"]), lo, hi, int(bad), int(so), float(ro)))
    return out


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    from oracle import oracle as O
    O.build()
    pool = mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2))
    import torch
    import polytope_amd as pa
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
    bad = npoly = 0
    t0 = time.time()
    for trial in range(trials):
        d = int(rng.choice([2, 3, 4, 5, 6, 8, 10]))
        m = int(rng.integers(65, 161))
        B = int(rng.integers(50, 600))
        fam = fams[int(rng.integers(0, len(fams)))]
        A, b, mrows = SL.make(rng, B, m, d, fam)
        if fam == "ragged":
            mrows = np.maximum(mrows, 1).astype(np.int32)
        At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mrows).to(dev)
        rd = pa.reduce_batch(At, bt, mt)
        ch = pa.cheby_ball_batch(At, bt, m=mt)
        # (plp_bbox_batch stops at 64 rows: the public bounding_box sends taller polytopes' 2 d LPs through the generic LP batch --
        # the LDS engine + the verifier; a sample of each batch goes that way)
        import polytope_amd.polytope as pc
        from polytope_amd import solvers
        solvers.default_solver = "hip"
        nbox = min(B, 12)
        boxes = [pc.bounding_box(pc.Polytope(A[k, :mrows[k]].copy(), b[k, :mrows[k]].copy(), normalize=False)) for k in range(nbox)]
        torch.cuda.synchronize()
        keep = rd["keep"].cpu().numpy().view(np.uint64).reshape(B, -1)
        flags, nlp, r = rd["flags"].cpu().numpy(), rd["nlp"].cpu().numpy(), rd["r"].cpu().numpy()
        cs, cr = ch["status"].cpu().numpy(), ch["r"].cpu().numpy()
        ref = []
        tasks = [(A[i:i + 8], b[i:i + 8], mrows[i:i + 8]) for i in range(0, B, 8)]
        for part in pool.imap(_task, tasks, chunksize=1):
            ref.extend(part)
        nr = nc = nb = 0
        first = None
        for k, (words, fl, nl, rr, lo, hi, bd, so, ro) in enumerate(ref):
            W = keep.shape[1]
            ok = tuple(int(w) for w in keep[k]) == tuple(words[:W]) and int(flags[k]) == fl and int(nlp[k]) == nl and \
                abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr))
            if not ok and fam == "dup" and tuple(int(w) for w in keep[k]) == tuple(words[:W]) and int(flags[k]) == fl and int(nlp[k]) == nl:
                ok = SL.highs_radius_agrees(A[k, :mrows[k]], b[k, :mrows[k]], r[k])
            if not ok and int(flags[k]) == fl and abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr)) and tuple(int(w) for w in keep[k]) == tuple(words[:W]) \
                    and SL.prefilter_tie(A[k, :mrows[k]], b[k, :mrows[k]]):
                ok = True
            if not ok:
                nr += 1
                first = first or ("reduce", k, [hex(int(w)) for w in keep[k]], [hex(w) for w in words[:W]], int(flags[k]), fl, int(nlp[k]), nl, r[k], rr)
            okc = int(cs[k]) == so and (so != 0 or abs(cr[k] - ro) <= 1e-9 * max(1.0, abs(ro)))
            if not okc:
                nc += 1
                first = first or ("cheby", k, int(cs[k]), so, cr[k], ro)
            if k < nbox and bd == 0:
                lb, ub = boxes[k][0].ravel(), boxes[k][1].ravel()
                okb = SL.box_equal(lb, ub, lo, hi, hair_unbounded_tol=(1e-8 if fam == "dup" else None))
                if not okb and fam == "dup":
                    okb = SL.box_equal(lb, ub, lo, hi, tol=SL.hair_tol(A[k, :mrows[k]]), hair_unbounded_tol=1e-8)
                if not okb:
                    nb += 1
                    first = first or ("bbox", k, lb, lo, ub, hi)
        npoly += B
        bad += nr + nc + nb
        print("trial %3d  d %2d m %3d B %4d  %-9s reduce bad %d  cheby bad %d  bbox bad %d   %s" % (trial, d, m, B, fam, nr, nc, nb, "" if first is None else first), flush=True)
    print("TALL SOAK %s: %d polytopes of 65..160 rows, %d mismatches, %.0f s" % ("FAILED" if bad else "OK", npoly, bad, time.time() - t0), flush=True)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
