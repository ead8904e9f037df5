#!/usr/bin/env python3
"""Soak of the one-LP-per-lane kernels (plp_reduce_lane.hip) on the GPU box, not part of the test suite: fused reduce and
stand-alone bounding boxes at (<= 32 rows, d <= 4) over every dispatch class (4 / 8 / 16 polytopes per wavefront, the mixed
launch, 17..32 rows, d = 4 forced and by size), on random, ragged, unbounded-allowed, duplicated / nearly duplicated,
rescaled and structured data -- EVERY polytope against the oracle (keep mask, flags, LP count exact; radius 1e-9), the
oracle on all host cores (workers forked before HIP exists in the process).
Usage: gpurun --timeout 1500 -- 'python scripts/soak_lane.py [trials] [seed]'"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _task(args):
    from oracle import oracle as O
    kind, A, b, m = args
    out = []
    for k in range(A.shape[0]):
        Ak, bk = A[k, :m[k]], b[k, :m[k]]
        if kind == "reduce":
            o = O.reduce(Ak, bk)
            out.append((int(o["mask"]), int(o["flags"]), int(o["nlp"]), float(o["r"])))
        else:
            lo, hi, bad = O.bounding_box(Ak, bk)
            so, ro, xo = O.cheby(Ak, bk)
            # (the ball's radius is known to 1e-9 of how far its centre lies: a sliver's centre 1e8 away is such a case)
            out.append((lo, hi, int(bad), int(so), float(ro), float(np.max(np.abs(xo))) if so == 0 else 1.0))
    return out


def oracle_all(pool, kind, A, b, m, chunk=64):
    B = A.shape[0]
    tasks = [(kind, A[i:i + chunk], b[i:i + chunk], m[i:i + chunk]) for i in range(0, B, chunk)]
    res = []
    for part in pool.imap(_task, tasks, chunksize=1):
        res.extend(part)
    return res


def highs_radius_agrees(Ak, bk, r_mine):
    """the Chebyshev LP (ref :1283-1288) by scipy / HiGHS: does its radius agree with r_mine to HiGHS's own tolerance?"""
    from scipy.optimize import linprog
    nrm = np.sqrt(np.sum(Ak * Ak, 1))
    c = np.zeros(Ak.shape[1] + 1)
    c[-1] = -1.0
    rs = linprog(c, np.hstack([Ak, nrm[:, None]]), bk, bounds=(None, None))
    return rs.status == 0 and abs(-rs.fun - r_mine) <= 1e-6 * max(1.0, abs(r_mine))


def box_equal(lb, ub, lo, hi, tol=1e-9, hair_unbounded_tol=None):
    """A box against the oracle's (round 6: both sides certified -- plp_verify.hpp / oracle lp_certify -- or re-solved in
    extended precision): +-inf in the same places, finite sides within `tol` of the box's EXTENT (the largest finite
    coordinate, at least 1: a side of 1.5 of a sliver that reaches 1e6 elsewhere is known to 1e-16 x 1e6, not to 1e-16).
    hair_unbounded_tol (family `dup` only): the tolerance where the polytope is UNBOUNDED (an infinite side).  With rows a hair
    apart a finite side of such a polytope is attained at vertices arbitrarily far out -- the hair's tilt, 1e-16, times the
    distance -- and two certified answers differ by the certificate's tolerance on a multiplier (1e-13) times how far apart
    their vertices lie: 4e-9 on a side of 3 in six of 3 M soaked polytopes (profiles/r06/soak_wide_*_73.log, trial 161)."""
    a, o = np.concatenate([lb, ub]), np.concatenate([lo, hi])
    fa, fo = np.isfinite(a), np.isfinite(o)
    if not np.array_equal(fa, fo) or not np.array_equal(a[~fa], o[~fo]):
        return False
    if hair_unbounded_tol is not None and not fo.all():
        tol = max(tol, hair_unbounded_tol)
    ext = max(1.0, float(np.max(np.abs(o[fo]), initial=1.0)))
    return bool(np.all(np.abs(a[fa] - o[fo]) <= tol * ext))


def hair_tol(Ak):
    """The tolerance (relative to the extent) a vertex that sits on two rows a hair apart is DEFINED to: the rows as stored carry
    half an ulp each, and planes an angle theta apart meet in a line that moves by that rounding divided by theta -- 1e-16 / 1e-9
    rad = 1e-7 of the extent.  theta = the smallest angle between two rows of the polytope (parallel or antiparallel); used by
    the soaks to CLASSIFY a box mismatch on family `dup` (counted separately), never to pass one silently."""
    nrm = np.sqrt((Ak * Ak).sum(1))
    An = Ak[nrm > 0] / nrm[nrm > 0][:, None]
    # (1 - |cos| loses the angle below 1e-8: the chord |a_i -+ a_j| keeps it; copies -- exact, or an ulp away -- define nothing)
    i, j = np.triu_indices(len(An), 1)
    ch = np.minimum(np.linalg.norm(An[i] - An[j], axis=1), np.linalg.norm(An[i] + An[j], axis=1))
    ch = ch[ch > 1e-13]     # (below: the same row to every LP code, this library's certificate included)
    return max(1e-9, 4e-16 / float(ch.min())) if ch.size else 1e-9


def prefilter_tie(Ak, bk, margin=1e-6):
    """reduce()'s bounding-box prefilter (ref :1131-1134) drops a row when  sum_k max(a_k, 0) (u_k - l_k) - (b - a.l) < -1e-4.
    True when some row of this polytope sits within `margin` (of the box's extent) of that threshold with the CERTIFIED box: which side it falls on
    is then decided by the last digits of the box LPs -- the reference's (HiGHS: 1e-7 feasibility tolerance), the oracle's
    dictionary simplex and the kernels' walk each have their own -- and with it the number of redundancy LPs issued (nlp);
    the kept rows do not depend on it.  Classified, not counted, by the soaks."""
    from oracle import oracle as O
    lo, hi, bad = O.bounding_box(Ak, bk)
    if bad or not (np.all(np.isfinite(lo)) and np.all(np.isfinite(hi))):
        return True
    val = ((Ak > 0) * Ak) @ (hi - lo) - (bk - Ak @ lo)
    ext = max(1.0, float(np.max(np.abs(np.concatenate([lo, hi])))))   # (a sliver that reaches 7e7: its box is known to 1e-9 of THAT)
    return bool(np.any(np.abs(val + 1e-4) < margin * ext))


def _classes(A, b, keep):
    """the kept rows as classes: rows that coincide to 1e-12 stand for one another (tests/test_oracle_golden.py: _twin_classes)"""
    nrm = np.sqrt((A * A).sum(1))
    An, bn = A / nrm[:, None], b / nrm
    rep = np.arange(len(b))
    for i in range(len(b)):
        for j in range(i):
            if np.abs(An[i] - An[j]).max() < 1e-12 and abs(bn[i] - bn[j]) < 1e-12 * max(1.0, abs(bn[i])):
                rep[i] = rep[j]
                break
    return np.unique(rep[np.asarray(keep, bool)])


def public_reduce_agrees(Ak, bk, oracle_mask):
    """RF_F1OPEN (flags & 32) in the fused kernel's answer and not in the oracle's: the kernel's Chebyshev LP ended "unbounded", at
    a limit, or at a centre outside the polytope where the oracle's did not (the two engines are restatements of one another, not
    bit-for-bit twins on rows a hair apart).  The flag MEANS "not answered here": polytope_amd.polytope.reduce re-examines such a
    polytope through the verified LPs.  So does this check: the product's public reduce() must keep the rows the oracle keeps."""
    import polytope_amd.polytope as pc
    from polytope_amd import solvers
    old = solvers.default_solver
    solvers.default_solver = "hip"
    try:
        q = pc.reduce(pc.Polytope(Ak.copy(), bk.copy(), normalize=False))
    finally:
        solvers.default_solver = old
    m = len(bk)
    want = np.array([(oracle_mask >> i) & 1 for i in range(m)], bool)
    if q.A.size == 0:
        return not want.any()
    nrm = np.sqrt((Ak * Ak).sum(1))
    An, bn = Ak / nrm[:, None], bk / nrm
    keep = np.zeros(m, bool)
    for a, bb in zip(q.A, q.b):
        dist = np.abs(An - a).max(1) + np.abs(bn - bb) / max(1.0, abs(bb))
        best = int(np.argmin(dist))
        if dist[best] > 1e-9:
            return False
        keep[best] = True
    return np.array_equal(_classes(Ak, bk, keep), _classes(Ak, bk, want))


def make(rng, B, m, d, fam):
    A = rng.standard_normal((B, m, d))
    A /= np.linalg.norm(A, axis=2, keepdims=True)
    b = 0.5 + rng.random((B, m))
    mrows = np.full(B, m, np.int32)
    if fam != "unbounded" and m >= 2 * d:
        A[:, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None]
        b[:, :2 * d] = rng.choice([1.5, 2.0, 3.0])
    if fam == "ragged":
        mrows = rng.integers(1, m + 1, B).astype(np.int32)
    elif fam == "dup" and m >= 4:
        # exact copies, copies with a shifted right-hand side, copies an ulp / 1e-9 / 1e-7 away
        for _ in range(max(1, m // 4)):
            i, j = rng.integers(0, m, 2)
            A[:, j] = A[:, i]
            b[:, j] = b[:, i] + rng.choice([0.0, 0.0, 1e-7, -1e-7, 0.1])
            eps = rng.choice([0.0, 1e-16, 1e-9, 1e-7, 1e-5])
            A[:, j] += eps * rng.standard_normal((B, d))
    elif fam == "scaled":
        s = np.exp(rng.uniform(-2.5, 2.5, (B, m)))
        A *= s[:, :, None]
        b *= s
    elif fam == "flat":
        # a slab of width 0 .. 3 abs_tol in a random direction on a part of the batch (empty / not full-dimensional)
        sel = rng.random(B) < 0.3
        n = rng.standard_normal((B, d))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        if m >= 2:
            A[sel, m - 1] = n[sel]
            A[sel, m - 2] = -n[sel]
            w = rng.choice([-1e-3, 0.0, 1e-7, 2.5e-7, 1e-3], B)
            b[sel, m - 1] = 0.2
            b[sel, m - 2] = -0.2 + w[sel]
    elif fam == "lattice":
        A = rng.integers(-2, 3, (B, m, d)).astype(float)
        z = np.abs(A).sum(2) == 0
        A[z, 0] = 1.0
        b = rng.integers(1, 5, (B, m)) * 0.5
        if m >= 2 * d:
            A[:, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None]
            b[:, :2 * d] = 2.0
    return np.ascontiguousarray(A), np.ascontiguousarray(b), mrows


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
    bad = 0
    npoly = 0
    n_oracle_off = 0   # answers on nearly duplicated rows where HiGHS sides with the kernel against the oracle
    n_cond = 0         # boxes that differ by less than the conditioning of rows a hair apart allows (hair_tol)
    n_tie = 0          # polytopes whose LP count alone differs, a row sitting on the prefilter's threshold (prefilter_tie)
    n_open = 0         # polytopes the fused kernel handed back (RF_F1OPEN) where the oracle's engine answered; public reduce() checked
    t0 = time.time()
    for trial in range(trials):
        d = int(rng.choice([1, 2, 3, 3, 3, 4, 4]))
        m = int(rng.integers(d + 1, 33))
        cls = trial % 6
        B = [int(rng.integers(1, 300)), int(rng.integers(2000, 9000)), int(rng.integers(15000, 30000)),
             int(rng.integers(41000, 60000)), int(rng.integers(300, 2000)), int(rng.integers(30001, 36000))][cls]
        if m > 16 or d == 4:
            B = min(B, 36000)
        fam = fams[int(rng.integers(0, len(fams)))]
        force = bool(rng.random() < 0.5)   # d = 4: small batches (plp_reduce_r.hip: PLP_REDUCE_LANE4_MINB*) go to the lane kernel only when asked to
        for k_ in ("PLP_REDUCE_LANE", "PLP_REDUCE_LANE_GS"):
            os.environ.pop(k_, None)
        if force:
            os.environ["PLP_REDUCE_LANE"] = "1"
            if rng.random() < 0.4:
                os.environ["PLP_REDUCE_LANE_GS"] = str(rng.choice([4, 8, 16]))
        A, b, mrows = make(rng, B, m, d, fam)
        At, bt, mt = torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mrows).to(dev)
        rd = pa.reduce_batch(At, bt, mt)
        torch.cuda.synchronize()
        keep = rd["keep"].cpu().numpy().view(np.uint64)
        flags = rd["flags"].cpu().numpy()
        nlp = rd["nlp"].cpu().numpy()
        r = rd["r"].cpu().numpy()
        ref = oracle_all(pool, "reduce", A, b, mrows)
        nb = 0
        first = None
        for k, (mk, fl, nl, rr) in enumerate(ref):
            ok = int(keep[k]) == mk and int(flags[k]) == fl and int(nlp[k]) == nl and \
                (abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr)) or (not np.isfinite(rr) and not np.isfinite(r[k])))
            if not ok and fam == "dup" and int(keep[k]) == mk and int(flags[k]) == fl and int(nlp[k]) == nl:
                # only the Chebyshev radius differs: on nearly duplicated rows the ORACLE's dictionary simplex can be the wrong
                # one (seed 103, trial 25: its ball sticks 0.046 out of the polytope).  HiGHS arbitrates, as for the boxes below.
                ok = highs_radius_agrees(A[k, :mrows[k]], b[k, :mrows[k]], r[k])
                n_oracle_off += int(ok)
            if not ok and int(keep[k]) == mk and int(flags[k]) == fl and abs(r[k] - rr) <= 1e-9 * max(1.0, abs(rr)) \
                    and prefilter_tie(A[k, :mrows[k]], b[k, :mrows[k]]):
                ok = True          # only the LP count differs, and a row sits on the prefilter's threshold (see prefilter_tie)
                n_tie += 1
            if not ok and (int(flags[k]) & 32) and not (fl & 32) and public_reduce_agrees(A[k, :mrows[k]], b[k, :mrows[k]], mk):
                ok = True          # the kernel handed the polytope back (RF_F1OPEN); the public reduce() keeps the oracle's rows
                n_open += 1
            if not ok:
                nb += 1
                first = first if first is not None else (k, hex(int(keep[k])), hex(mk), int(flags[k]), fl, int(nlp[k]), nl, r[k], rr)
        # bounding boxes of a part of the batch (d <= 3: the lane form; d = 4: the lane-group kernels)
        nbb = 0
        nq = min(B, 3000)
        bb = pa.bbox_batch(At[:nq], bt[:nq], mt[:nq])
        if bb is not None:
            torch.cuda.synchronize()
            st = bb["status"].cpu().numpy()
            lb, ub = bb["lb"].cpu().numpy(), bb["ub"].cpu().numpy()
            refb = oracle_all(pool, "bbox", A[:nq], b[:nq], mrows[:nq])
            for k, (lo, hi, bd, so, ro, _xn) in enumerate(refb):
                if st[k] != 0:
                    continue
                okb = bd == 0 and box_equal(lb[k], ub[k], lo, hi, hair_unbounded_tol=(1e-8 if fam == "dup" else None))
                if not okb and bd == 0 and fam == "dup" and box_equal(lb[k], ub[k], lo, hi, tol=hair_tol(A[k, :mrows[k]]), hair_unbounded_tol=1e-8):
                    okb = True       # within what two rows a hair apart define (hair_tol): classified, counted
                    n_cond += 1
                if not okb:
                    nbb += 1
                    first = first if first is not None else ("bbox", k, lb[k], lo, ub[k], hi, bd)
        npoly += B
        bad += nb + nbb
        print("trial %3d  d %d m %2d B %6d  %-9s force %d gs %-2s  reduce bad %d  bbox bad %d   %s" % (
            trial, d, m, B, fam, force, os.environ.get("PLP_REDUCE_LANE_GS", "-"), nb, nbb, "" if first is None else first),
            flush=True)
    print("LANE SOAK %s: %d polytopes, %d mismatches, %.0f s  (radii on nearly duplicated rows where HiGHS sides with the fused "
          "kernel against the oracle's raw engine: %d; LP counts that differ on a prefilter tie: %d; boxes within the conditioning of "
          "rows a hair apart: %d; handed back by the fused kernel (RF_F1OPEN) and right through the public reduce(): %d)" % (
              "FAILED" if bad else "OK", npoly, bad, time.time() - t0, n_oracle_off, n_tie, n_cond, n_open), flush=True)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
