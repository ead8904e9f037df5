"""CPU model: pivots of the 2d bounding-box LPs from the Chebyshev centre (fresh dictionary each) against a chain that
keeps the dictionary and swaps the cost row.  Same pivot rules as the engines (Dantzig on |c| for free columns, -c for
slacks; ratio test lowest row on ties)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from polytope_amd.synth import random_hpolytopes
from scipy.optimize import linprog

TOL_D = 1e-9; TOL_PIV = 1e-9

class Dict:
    def __init__(s, A, beta):
        s.T = A.copy(); s.beta = beta.copy(); m, n = A.shape
        s.m, s.n = m, n
        s.rv = list(range(n, n + m)); s.cv = list(range(n)); s.rneg = [False] * m; s.cneg = [False] * n
        s.ract = [True] * m; s.cfree = [True] * n
        s.cost = np.zeros(n); s.negz = 0.0; s.piv = 0
    def set_cost_coord(s, k, sigma):
        # minimise sigma * y_k
        s.cost[:] = 0; s.negz = 0.0
        if k in s.cv:
            j = s.cv.index(k); s.cost[j] = sigma * (-1 if s.cneg[j] else 1)
        else:
            i = s.rv.index(k); sg = -1.0 if s.rneg[i] else 1.0
            s.cost[:] = -sigma * sg * s.T[i]; s.negz = -sigma * sg * s.beta[i]
    def run(s):
        while True:
            e = -1; best = TOL_D
            for j in range(s.n):
                key = abs(s.cost[j]) if s.cfree[j] else -s.cost[j]
                if key > best: best = key; e = j
            if e < 0: return 0
            a = s.T[:, e].copy(); flip = s.cost[e] > 0
            if flip: a = -a
            r = -1; bn = np.inf; an = 1.0
            for i in range(s.m):
                if s.ract[i] and a[i] > TOL_PIV:
                    bi = max(s.beta[i], 0.0)
                    if bi * an < bn * a[i]: r = i; bn = bi; an = a[i]
            if r < 0: return 3
            p = 1.0 / an
            rho = s.T[r] * p; rhob = s.beta[r] * p
            fc = -best
            s.cost = s.cost - fc * rho; s.negz -= fc * rhob
            for i in range(s.m):
                if i == r: continue
                f = a[i]
                s.T[i] = s.T[i] - f * rho; s.beta[i] -= f * rhob; s.T[i, e] = -(f * p)
            s.cost[e] = -(fc * p)
            rho[e] = p
            s.T[r] = rho; s.beta[r] = rhob
            vin, vout = s.cv[e], s.rv[r]
            eneg = s.cneg[e] ^ flip
            s.cv[e] = vout; s.cneg[e] = s.rneg[r]
            s.rv[r] = vin; s.rneg[r] = eneg
            s.ract[r] = not s.cfree[e]
            s.cfree[e] = False
            s.piv += 1

def centre(A, b):
    m, d = A.shape
    G = np.hstack([A, np.linalg.norm(A, axis=1)[:, None]])
    c = np.zeros(d + 1); c[d] = -1
    res = linprog(c, A_ub=G, b_ub=b, bounds=[(None, None)] * (d + 1), method="highs")
    return res.x[:d], res.x[d]

def orders(d):
    yield "min0 max0 min1 max1 ..", [(k, s) for k in range(d) for s in (1, -1)]
    yield "all min then all max", [(k, 1) for k in range(d)] + [(k, -1) for k in range(d)]
    yield "min0..min(d-1) max(d-1)..max0", [(k, 1) for k in range(d)] + [(k, -1) for k in reversed(range(d))]

for (m, d, B) in [(16, 3, 300), (32, 6, 150), (64, 8, 60), (24, 5, 150)]:
    A, b = random_hpolytopes(B, m, d, seed=1, stream=0)
    fresh = []; chain = {}
    maxdiff = 0.0
    for p in range(B):
        xc, r = centre(A[p], b[p])
        if r <= 1e-7: continue
        beta = np.maximum(b[p] - A[p] @ xc, 0)
        vals = {}
        for k in range(d):
            for s in (1, -1):
                D = Dict(A[p], beta); D.set_cost_coord(k, s); st = D.run(); fresh.append(D.piv)
                vals[(k, s)] = (st, xc[k] - s * D.negz)
        for name, order in orders(d):
            D = Dict(A[p], beta); cnt = []
            for (k, s) in order:
                p0 = D.piv; D.set_cost_coord(k, s); st = D.run(); cnt.append(D.piv - p0)
                v = xc[k] - s * D.negz
                if st == 0 and vals[(k, s)][0] == 0: maxdiff = max(maxdiff, abs(v - vals[(k, s)][1]))
                assert st == vals[(k, s)][0]
            chain.setdefault(name, []).append(cnt)
    print("(%d,%d): fresh from the centre %.2f pivots per LP (max %d)" % (m, d, np.mean(fresh), max(fresh)))
    for name, c in chain.items():
        c = np.array(c)
        print("   chain [%s]: %.2f per LP (first LP %.2f, others %.2f, max %d); max |value diff| %.1e" % (name, c.mean(), c[:, 0].mean(), c[:, 1:].mean(), c.max(), maxdiff))
