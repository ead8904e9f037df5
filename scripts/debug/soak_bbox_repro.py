"""The bounding boxes soak_lane.py flagged (nearly duplicated rows): the values of both box kernels next to the oracle's."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as S


def data(wanted, seed=5):
    rng = np.random.default_rng(seed)
    fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
    out = {}
    for trial in range(max(wanted) + 1):
        d = int(rng.choice([1, 2, 3, 3, 3, 4, 4])); m = int(rng.integers(d + 1, 33)); cls = trial % 6
        B = [int(rng.integers(1, 300)), int(rng.integers(2000, 9000)), int(rng.integers(15000, 30000)),
             int(rng.integers(41000, 60000)), int(rng.integers(300, 2000)), int(rng.integers(30001, 36000))][cls]
        if m > 16 or d == 4: B = min(B, 36000)
        fam = fams[int(rng.integers(0, len(fams)))]
        if bool(rng.random() < 0.5):
            if rng.random() < 0.4: rng.choice([4, 8, 16])
        A, b, mrows = S.make(rng, B, m, d, fam)
        if trial in wanted: out[trial] = (A, b, mrows)
    return out


if __name__ == "__main__":
    import polytope_amd as pa
    from oracle import oracle as O
    O.build()
    np.set_printoptions(precision=17, linewidth=220)
    for trial, (A, b, mr) in data([4, 8, 34]).items():
        nq = min(A.shape[0], 3000)
        outs = {}
        for lane in ("1", "0"):
            os.environ["PLP_BBOX_LANE"] = lane
            outs[lane] = pa.bbox_batch(A[:nq], b[:nq], mr[:nq])
        shown = 0
        for k in range(nq):
            lo, hi, bad = O.bounding_box(A[k, :mr[k]], b[k, :mr[k]])
            for lane, bb in outs.items():
                if bb["status"][k] != 0: continue
                e = max(np.abs(np.where(np.isfinite(lo), bb["lb"][k] - lo, 0)).max(), np.abs(np.where(np.isfinite(hi), bb["ub"][k] - hi, 0)).max())
                if e > 1e-9 and shown < 6:
                    shown += 1
                    print("trial", trial, "poly", k, "lane", lane, "err %.3e" % e, "status other", outs["0" if lane == "1" else "1"]["status"][k])
                    print("  hip lb", bb["lb"][k], "ub", bb["ub"][k]); print("  ora lb", lo, "ub", hi)
