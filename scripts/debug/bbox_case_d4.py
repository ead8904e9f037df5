"""One (32,4) polytope with nearly duplicated rows on which bbox_r_kernel<4> returned a corner outside a row (soak_lane.py 90 7,
trial 5, polytope 2853): the fused box against the generic LPs, the oracle and HiGHS, in the engines' variants."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "debug"))
import soak_bbox_repro as R

if __name__ == "__main__":
    np.set_printoptions(precision=12, linewidth=220)
    A, b, mr = R.data([5], seed=7)[5]
    k = 2853
    Ak, bk = A[k, :mr[k]], b[k, :mr[k]]
    np.savez(os.path.join(ROOT, "gpurun_out", "bbox_case_d4.npz"), A=Ak, b=bk)
    from scipy.optimize import linprog
    from oracle import oracle as O
    O.build()
    lo, hi, bad = O.bounding_box(Ak, bk)
    print("oracle", lo, hi, bad)
    hl, hh = [], []
    for i in range(4):
        for s, dst in ((1, hl), (-1, hh)):
            c = np.zeros(4); c[i] = s
            r = linprog(c, Ak, bk, bounds=(None, None))
            dst.append(r.x[i] if r.status == 0 else np.nan)
    print("highs ", np.array(hl), np.array(hh))
    import polytope_amd as pa
    for env in ({}, {"PLP_BBOX_SPLIT": "1"}, {"PLP_CHEBY_RETRY_ALL": "1"}, {"PLP_BBOX_LAZY": "0"}):
        for k_, v in env.items(): os.environ[k_] = v
        bb = pa.bbox_batch(Ak[None], bk[None])
        print("fused", env, bb["status"], bb["lb"][0], bb["ub"][0])
        # in a batch with neighbours (the lane group shares a wavefront with 15 others)
        nb = min(A.shape[0], k + 40)
        bb = pa.bbox_batch(A[k - 24:nb], b[k - 24:nb], mr[k - 24:nb])
        print("  in batch", bb["status"][24], bb["lb"][24], bb["ub"][24])
        for k_ in env: os.environ.pop(k_)
    ch = pa.cheby_ball_batch(Ak[None], bk[None]); print("cheby", ch["status"], ch["r"], ch["xc"])
    print("oracle cheby", O.cheby(Ak, bk))
    c = np.vstack([np.eye(4), -np.eye(4)])
    lp = pa.lpsolve_batch(c, np.repeat(Ak[None], 8, 0), np.repeat(bk[None], 8, 0))
    print("generic LPs status", lp["status"], "x_k", [lp["x"][j][j % 4] for j in range(8)])
    # rows that the fused corner violates
    viol = Ak[:, 0] * 0
    print("rows nearly duplicated (cos > 1 - 1e-8):", [(i, j) for i in range(len(bk)) for j in range(i) if Ak[i] @ Ak[j] / np.linalg.norm(Ak[i]) / np.linalg.norm(Ak[j]) > 1 - 1e-8])
