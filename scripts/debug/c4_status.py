"""Debug: statuses of every Chebyshev LP region_diff issues at C4 on the hip backend; failing LPs re-solved by scipy."""
import sys, os, itertools, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.optimize import linprog
import polytope_amd.polytope as pc
from polytope_amd import solvers, batch
from conftest import load_golden
solvers.default_solver = "hip"
g = load_golden("g12_config4.npz")
shape = tuple(int(v) for v in g["c4_shape"])
cells = [pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]) for idx in itertools.product(*[range(n) for n in shape])]
P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
orig = batch.cheby_ball_batch
log = {"calls": 0, "lps": 0, "bad": [], "near": 0, "mism": 0}
def spy(A, b, m=None):
    res = orig(A, b, m=m)
    log["calls"] += 1
    B = A.shape[0]
    log["lps"] += B
    ms = np.full(B, A.shape[1]) if m is None else np.asarray(m)
    st = np.asarray(res["status"]); r = np.asarray(res["r"])
    for k in np.nonzero(st != 0)[0][:50]:
        mk = int(ms[k]); d = A.shape[2]
        G = np.c_[A[k, :mk], np.sqrt(np.sum(A[k, :mk] ** 2, 1))]
        sp = linprog(np.r_[np.zeros(d), -1.0], G, b[k, :mk], None, None, bounds=(None, None))
        log["bad"].append((int(st[k]), mk, int(sp.status), None if sp.x is None else float(sp.x[-1])))
    # sample a few optimal ones against scipy
    rng = np.random.default_rng(log["calls"])
    for k in rng.choice(B, size=min(B, 2), replace=False):
        if st[k] != 0: continue
        mk = int(ms[k]); d = A.shape[2]
        G = np.c_[A[k, :mk], np.sqrt(np.sum(A[k, :mk] ** 2, 1))]
        sp = linprog(np.r_[np.zeros(d), -1.0], G, b[k, :mk], None, None, bounds=(None, None))
        if sp.status != 0 or abs(sp.x[-1] - r[k]) > 1e-9:
            log["mism"] += 1
            print("MISMATCH m", mk, "hip r", r[k], "scipy", sp.status, None if sp.x is None else sp.x[-1], flush=True)
        if abs(r[k] - 1e-7) < 1e-9: log["near"] += 1
    return res
batch.cheby_ball_batch = spy
D = pc.region_diff(P.copy(), pc.Region(cells[:500]))
print("pieces", len(D), "want", int(g["c4_diff_n"]))
print("calls", log["calls"], "lps", log["lps"], "mismatch", log["mism"], "near", log["near"])
from collections import Counter
print("bad statuses:", Counter((a, c) for a, _, c, _ in log["bad"]), "examples", log["bad"][:10])
