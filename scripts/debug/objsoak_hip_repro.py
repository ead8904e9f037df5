"""One trial of scripts/soak_objects_hip.py in detail.   python scripts/debug/objsoak_hip_repro.py <seed> <trial> <op>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import logging; logging.disable(logging.CRITICAL)
import polytope_amd as pc
from polytope_amd import solvers
from soak_objects_hip import rand_poly, pieces
seed, want, op = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed)
for trial in range(want + 1):
    d = int(rng.choice([1, 2, 2, 3, 3, 4, 4]))
    kinds = [str(rng.choice(["box", "boxed", "boxed", "free"])) for _ in range(3)]
    data = [rand_poly(rng, d, k) for k in kinds]
    if trial % 4 == 0:
        A0, b0 = data[0]
        n = rng.standard_normal(d); n /= np.linalg.norm(n)
        off = float(n @ rng.uniform(-0.2, 0.2, d))
        data[1] = (np.vstack([A0, n]), np.hstack([b0, off]))
        data[2] = (np.vstack([A0, -n]), np.hstack([b0, -off]))
ops = {"mldivide(Region, P)": lambda P: pc.mldivide(pc.Region([P[0], P[1]]), P[2]),
       "Region.diff": lambda P: pc.Region([P[1], P[2]]).diff(P[0]),
       "mldivide(P, Region)": lambda P: pc.mldivide(P[0], pc.Region([P[1], P[2]])),
       "mldivide": lambda P: pc.mldivide(P[0], P[1])}
np.set_printoptions(linewidth=200, precision=6, suppress=True)
out = {}
for backend in ("scipy", "hip"):
    solvers.default_solver = backend
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    np.random.seed(want)
    out[backend] = pieces(ops[op](P))
print("d", d, kinds, "pieces", len(out["scipy"]), len(out["hip"]))
def key(p):
    M = np.round(np.c_[p.A, p.b], 6) + 0.0
    return M[np.lexsort(M.T[::-1])]
for k in range(max(len(out["scipy"]), len(out["hip"]))):
    a = out["scipy"][k] if k < len(out["scipy"]) else None
    b = out["hip"][k] if k < len(out["hip"]) else None
    if a is None or b is None:
        print("piece", k, "only on", "hip" if a is None else "scipy"); continue
    same_order = a.A.shape == b.A.shape and np.allclose(np.c_[a.A, a.b], np.c_[b.A, b.b], atol=1e-9)
    same_set = a.A.shape == b.A.shape and np.allclose(key(a), key(b), atol=1e-6)
    print("piece", k, "rows", a.A.shape[0], b.A.shape[0], "same order", same_order, "same set", same_set, "r", float(pc.cheby_ball(a)[0]), float(pc.cheby_ball(b)[0]))
# does the set of pieces agree up to order?
ks = sorted([key(p).tobytes() for p in out["scipy"]]); kh = sorted([key(p).tobytes() for p in out["hip"]])
print("same pieces as a set:", ks == kh)
if len(sys.argv) > 4 and sys.argv[4] == "env":
    # which primitive orders rows differently?  envelope / reduce / is_convex of every pair of the pieces of the two differences
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    parts = pieces(P[0].diff(P[2])) + pieces(P[1].diff(P[2]))
    print("parts", len(parts))
    import itertools
    for i, j in itertools.permutations(range(len(parts)), 2):
        res = {}
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            R = pc.Region([parts[i].copy(), parts[j].copy()])
            e = pc.envelope(R)
            res[backend] = (np.c_[e.A, e.b] if e.A.size else np.zeros((0, d + 1)), pc.is_convex(pc.Region([parts[i].copy(), parts[j].copy()]))[0])
            if e.A.size:
                q = pc.reduce(e)
                res[backend] += (np.c_[q.A, q.b],)
            else:
                res[backend] += (np.zeros((0, d + 1)),)
        a, b_ = res["scipy"], res["hip"]
        ok_env = a[0].shape == b_[0].shape and np.allclose(a[0], b_[0], atol=1e-9)
        ok_red = a[2].shape == b_[2].shape and np.allclose(a[2], b_[2], atol=1e-9)
        if not ok_env or not ok_red or bool(a[1]) != bool(b_[1]):
            print("pair", i, j, "envelope same:", ok_env, a[0].shape, b_[0].shape, " is_convex", a[1], b_[1], " reduce(envelope) same:", ok_red, a[2].shape, b_[2].shape)
            if not ok_env and a[0].shape == b_[0].shape:
                print("scipy env\n", a[0], "\nhip env\n", b_[0])
if len(sys.argv) > 4 and sys.argv[4] == "parts":
    for i in (0, 1):
        res = {}
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
            res[backend] = pieces(P[i].diff(P[2]))
        print("P[%d] \\ P[2]:" % i, len(res["scipy"]), len(res["hip"]), [bool(a.A.shape == b.A.shape and np.allclose(np.c_[a.A, a.b], np.c_[b.A, b.b], atol=1e-9)) for a, b in zip(res["scipy"], res["hip"])])
    # the chain itself, step by step
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        out = pc.Region()
        for poly in (P[0], P[1]):
            rest = pc.mldivide(poly, P[2])
            out = pc.union(out, rest, check_convex=True)
            print(backend, "after member: pieces", [(p.A.shape[0], round(float(pc.cheby_ball(p)[0]), 6)) for p in pieces(out)])
if len(sys.argv) > 4 and sys.argv[4] == "rows":
    keep = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        out = pc.Region()
        for step, poly in enumerate((P[0], P[1])):
            rest = pc.mldivide(poly, P[2])
            out = pc.union(out, rest, check_convex=True)
            keep[(backend, step)] = np.c_[pieces(out)[0].A, pieces(out)[0].b]
    for step in (0, 1):
        a, b_ = keep[("scipy", step)], keep[("hip", step)]
        print("step", step, "piece 0 same order:", a.shape == b_.shape and np.allclose(a, b_, atol=1e-9))
        if not (a.shape == b_.shape and np.allclose(a, b_, atol=1e-9)):
            print("scipy\n", a, "\nhip\n", b_)
if len(sys.argv) > 4 and sys.argv[4] == "single":
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    parts = pieces(P[0].diff(P[2])) + pieces(P[1].diff(P[2]))
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        for k, p in enumerate(parts):
            q = p.copy()
            e = pc.envelope(pc.Region([q]))
            h = pc.reduce(e)
            h2 = pc.reduce(h)
            f = lambda x: np.c_[x.A, x.b]
            print(backend, "part", k, "rows", p.A.shape[0], "envelope same order:", e.A.shape == p.A.shape and np.allclose(f(e), f(p), atol=1e-12),
                  "reduce same order:", h.A.shape == p.A.shape and np.allclose(f(h), f(p), atol=1e-12), "again:", h2.A.shape == p.A.shape and np.allclose(f(h2), f(p), atol=1e-12))
        # the whole second union with the memos cleared
        import polytope_amd.polytope as pp
        pp._hull_memo.clear(); pp._convex_memo.clear()
        a = pc.Region([x.copy() for x in parts[:3]]); b_ = pc.Region([x.copy() for x in parts[3:]])
        U = pc.union(a, b_, check_convex=True)
        print(backend, "union piece 0 same order as part 0:", np.allclose(np.c_[pieces(U)[0].A, pieces(U)[0].b], np.c_[parts[0].A, parts[0].b], atol=1e-12) if pieces(U)[0].A.shape == parts[0].A.shape else "shape")
if len(sys.argv) > 4 and sys.argv[4] == "outer":
    import itertools
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    parts = pieces(P[0].diff(P[2])) + pieces(P[1].diff(P[2]))
    f = lambda x: np.c_[x.A, x.b]
    for i, j in itertools.permutations(range(len(parts)), 2):
        rows = {}
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            cv, outer = pc.is_convex(pc.Region([parts[i].copy(), parts[j].copy()]))
            env = pc.envelope(pc.Region([parts[i].copy(), parts[j].copy()]))
            rows[backend] = (bool(cv), None if outer is None else f(outer), f(env) if env.A.size else None)
        for backend in ("scipy", "hip"):
            cv, o, e = rows[backend]
            if cv and o is not None and e is not None and not (o.shape == e.shape and np.allclose(o, e, atol=1e-12)):
                print(backend, "pair", i, j, "convex: the envelope is_convex returns differs from envelope():\n", o, "\n", e)
        if rows["scipy"][0] and rows["hip"][0]:
            o1, o2 = rows["scipy"][1], rows["hip"][1]
            print("pair", i, j, "convex on both; is_convex's envelope same order across backends:", o1.shape == o2.shape and np.allclose(o1, o2, atol=1e-12))
if len(sys.argv) > 4 and sys.argv[4] == "trace":
    import polytope_amd.polytope as pp
    solvers.default_solver = "hip"
    target = np.array([0.974584, -0.224023])
    def has(p): return p.A.size and np.any(np.all(np.abs(p.A - target) < 1e-5, axis=1))
    def pos(p): return int(np.flatnonzero(np.all(np.abs(p.A - target) < 1e-5, axis=1))[0]) if has(p) else -1
    o_env, o_red, o_rm = pp.envelope, pp.reduce, pp._reduce_many
    def env(reg, abs_tol=pp.ABS_TOL):
        out = o_env(reg, abs_tol)
        print("  envelope(", [ (m.A.shape[0], pos(m)) for m in reg.list_poly], ") ->", out.A.shape[0], pos(out)); return out
    def red(p, *a, **k):
        out = o_red(p, *a, **k)
        if isinstance(p, pp.Polytope): print("  reduce(", p.A.shape[0], pos(p), "minrep", p.minrep, ") ->", out.A.shape[0], pos(out))
        return out
    def rm(polys, *a, **k):
        out = o_rm(polys, *a, **k)
        print("  _reduce_many(", [(p.A.shape[0], pos(p)) for p in polys], ") ->", [(q.A.shape[0], pos(q)) if q is not None else None for q in out]); return out
    pp.envelope, pp.reduce, pp._reduce_many = env, red, rm
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    out = pc.Region()
    for step, poly in enumerate((P[0], P[1])):
        print("step", step)
        rest = pc.mldivide(poly, P[2])
        print(" rest pieces", [(p.A.shape[0], pos(p)) for p in pieces(rest)])
        out = pc.union(out, rest, check_convex=True)
        print(" out pieces", [(p.A.shape[0], pos(p)) for p in pieces(out)])
if len(sys.argv) > 4 and sys.argv[4] == "isect":
    f = lambda x: np.c_[x.A, x.b]
    res = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        I = pc.intersect(a, b_)
        D1 = b_.diff(a); D2 = a.diff(b_)
        res[backend] = [pieces(I), pieces(D1), pieces(D2)]
    for nm, k in (("intersect(a, b)", 0), ("b.diff(a)", 1), ("a.diff(b)", 2)):
        x, y = res["scipy"][k], res["hip"][k]
        print(nm, "pieces", len(x), len(y), [bool(p.A.shape == q.A.shape and np.allclose(f(p), f(q), atol=1e-9)) for p, q in zip(x, y)])
        if len(x) == len(y):
            for p, q in zip(x, y):
                if not (p.A.shape == q.A.shape and np.allclose(f(p), f(q), atol=1e-9)):
                    print(" scipy\n", f(p), "\n hip\n", f(q))
if len(sys.argv) > 4 and sys.argv[4] == "groups":
    import polytope_amd.polytope as pp
    f = lambda x: np.c_[x.A, x.b]
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        o_ic, o_cnc = pp.is_convex, pp._clearly_not_convex
        tag = lambda p: "%d:%.4f" % (p.A.shape[0], float(pc.cheby_ball(p)[0]))
        def ic(reg, abs_tol=pp.ABS_TOL):
            out = o_ic(reg, abs_tol); print("   is_convex", [tag(p) for p in reg.list_poly], "->", bool(out[0])); return out
        def cnc(group):
            out = o_cnc(group); print("   witness ", [tag(p) for p in group], "-> not convex" if out else "-> no witness"); return out
        pp.is_convex, pp._clearly_not_convex = ic, cnc
        print(backend)
        U = pc.union(a, b_, check_convex=True)
        pp.is_convex, pp._clearly_not_convex = o_ic, o_cnc
        print("  result", [tag(p) for p in pieces(U)])
if len(sys.argv) > 4 and sys.argv[4] == "memo":
    import polytope_amd.polytope as pp
    f = lambda x: np.c_[x.A, x.b]
    def run(backend):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        np.random.seed(want)
        return pieces(ops[op](P))
    def cmp(x, y): return [bool(p.A.shape == q.A.shape and np.allclose(f(p), f(q), atol=1e-9)) for p, q in zip(x, y)]
    pp._hull_memo.clear(); pp._convex_memo.clear()
    h1 = run("hip"); pp._hull_memo.clear(); pp._convex_memo.clear()
    s1 = run("scipy"); pp._hull_memo.clear(); pp._convex_memo.clear()
    print("memos cleared between runs: hip vs scipy", cmp(h1, s1))
    s2 = run("scipy"); h2 = run("hip")
    print("scipy then hip, memos kept:           ", cmp(h2, s2), " hip(kept) vs hip(fresh)", cmp(h2, h1))
    pp._hull_memo.clear(); pp._convex_memo.clear()
    h3 = run("hip"); h4 = run("hip")
    print("hip twice, memos kept: second vs first", cmp(h4, h3))
if len(sys.argv) > 4 and sys.argv[4] == "tie":
    import polytope_amd.polytope as pp
    from oracle import oracle as O
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
    lst = pieces(pc.intersect(a, b_)) + pieces(b_.diff(a)) + pieces(a.diff(b_))
    import itertools
    for i, j in itertools.permutations(range(len(lst)), 2):
        cv, outer = pc.is_convex(pc.Region([lst[i].copy(), lst[j].copy()]))
        if not cv: continue
        env = pc.envelope(pc.Region([lst[i].copy(), lst[j].copy()]))
        Ae, be = env.A.copy(), env.b.copy()
        o = O.reduce(Ae, be)
        masks = {"oracle": int(o["mask"])}
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            R = pc.reduce(pc.Polytope(Ae.copy(), be.copy(), normalize=False))
            m = 0
            for row, bb in zip(R.A, R.b):
                # the position of each kept row in the envelope: FIRST unused exact match
                for k in range(len(be)):
                    if not (m >> k) & 1 and np.allclose(Ae[k], row, atol=1e-12) and abs(be[k] - bb) < 1e-12:
                        cand = k; break
                # several envelope rows can match: report all candidates instead
            kept = [[k for k in range(len(be)) if np.allclose(Ae[k], row, atol=1e-12) and abs(be[k] - bb) < 1e-12] for row, bb in zip(R.A, R.b)]
            masks[backend] = kept
        solvers.default_solver = "scipy"
        print("pair", i, j, "envelope rows", len(be), "oracle keep", [k for k in range(len(be)) if (masks["oracle"] >> k) & 1])
        print("   scipy kept rows (candidate positions):", masks["scipy"])
        print("   hip   kept rows (candidate positions):", masks["hip"])
        nrm = np.sqrt((Ae * Ae).sum(1))
        print("   normalised b:", ["%.17g" % v for v in be / nrm])
if len(sys.argv) > 4 and sys.argv[4] == "order":
    f = lambda x: np.c_[x.A, x.b]
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
    lst = pieces(pc.intersect(a, b_)) + pieces(b_.diff(a)) + pieces(a.diff(b_))
    print("lst", [(p.A.shape[0], round(float(pc.cheby_ball(p)[0]), 4)) for p in lst])
    for (i, j) in ((0, 5), (5, 0)):
        for backend in ("scipy", "hip"):
            solvers.default_solver = backend
            R = pc.Region([lst[i].copy(), lst[j].copy()])
            cv, outer = pc.is_convex(R)
            print("members", (i, j), backend, "convex", bool(cv), "envelope rows", None if outer is None else outer.A.shape[0])
            if outer is not None:
                h = pc.reduce(pc.reduce(outer)); print(f(h))
if len(sys.argv) > 4 and sys.argv[4] == "iso":
    import polytope_amd.polytope as pp
    f = lambda x: np.c_[x.A, x.b]
    res = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        res[backend, "iso"] = pieces(pc.union(a, b_, check_convex=True))
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        out = pc.Region()
        out = pc.union(out, pc.mldivide(P[0], P[2]), check_convex=True)
        res[backend, "chain"] = pieces(pc.union(out, pc.mldivide(P[1], P[2]), check_convex=True))
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        res[backend, "op"] = pieces(pc.mldivide(pc.Region([P[0], P[1]]), P[2]))
    cmp = lambda x, y: [bool(p.A.shape == q.A.shape and np.allclose(f(p), f(q), atol=1e-9)) for p, q in zip(x, y)]
    for k in ("iso", "chain", "op"):
        print(k, "hip vs scipy", cmp(res["hip", k], res["scipy", k]))
    print("hip: op vs chain", cmp(res["hip", "op"], res["hip", "chain"]), " chain vs iso", cmp(res["hip", "chain"], res["hip", "iso"]))
if len(sys.argv) > 4 and sys.argv[4] == "trace2":
    import polytope_amd.polytope as pp
    fp = lambda p: "[" + " ".join("%.3f" % v for v in p.A[:, 0]) + "]" if p.A.size else "[]"
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        o_env, o_red, o_ic = pp.envelope, pp.reduce, pp.is_convex
        def env(reg, abs_tol=pp.ABS_TOL):
            out = o_env(reg, abs_tol); print("  envelope", [fp(m) for m in reg.list_poly], "->", fp(out)); return out
        def red(p, *a_, **k):
            out = o_red(p, *a_, **k)
            if isinstance(p, pp.Polytope): print("  reduce", fp(p), "minrep", p.minrep, "->", fp(out))
            return out
        pp.envelope, pp.reduce = env, red
        print(backend)
        U = pc.union(a, b_, check_convex=True)
        pp.envelope, pp.reduce = o_env, o_red
        print("  result piece 0", fp(pieces(U)[0]))
if len(sys.argv) > 4 and sys.argv[4] == "bits":
    import polytope_amd.polytope as pp
    from oracle import oracle as O
    caught = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        o_red = pp.reduce
        def red(p, *a_, **k):
            if isinstance(p, pp.Polytope) and p.A.shape[0] == 8 and not p.minrep:
                caught.setdefault(backend, (p.A.copy(), p.b.copy()))
            return o_red(p, *a_, **k)
        pp.reduce = red
        pc.union(a, b_, check_convex=True)
        pp.reduce = o_red
    (As, bs), (Ah, bh) = caught["scipy"], caught["hip"]
    print("8-row stack: A bits equal", np.array_equal(As, Ah), "b bits equal", np.array_equal(bs, bh), "max diff", np.abs(As - Ah).max(), np.abs(bs - bh).max())
    np.set_printoptions(precision=17, linewidth=200)
    an = 1 / np.sqrt(np.sum(Ah.T ** 2, 0))
    print("b*an (reference formula) on hip's input:", bh * an)
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        R = pc.reduce(pc.Polytope(Ah.copy(), bh.copy(), normalize=False))
        print(backend, "reduce of hip's input keeps first coeffs", np.round(R.A[:, 0], 3))
    o = O.reduce(Ah, bh); print("oracle mask", bin(int(o["mask"])))
    np.savez("gpurun_out/tie8.npz", A=Ah, b=bh)
if len(sys.argv) > 4 and sys.argv[4] == "ulp":
    res = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[0], P[2]); b_ = pc.mldivide(P[1], P[2])
        I = pc.intersect(a, b_); D1 = b_.diff(a); D2 = a.diff(b_)
        res[backend] = dict(P0=[P[0]], P2=[P[2]], a=pieces(a), b=pieces(b_), I=pieces(I), D1=pieces(D1), D2=pieces(D2))
    for k in ("P0", "P2", "a", "b", "I", "D1", "D2"):
        x, y = res["scipy"][k], res["hip"][k]
        print(k, [(bool(np.array_equal(p.A, q.A)), bool(np.array_equal(p.b, q.b)), float(np.abs(p.b - q.b).max()) if p.b.shape == q.b.shape else None) for p, q in zip(x, y)])
if len(sys.argv) > 4 and sys.argv[4] == "rc":
    import polytope_amd.polytope as pp
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        Rc = pp._radii_stacked(P[0].copy(), [P[1], P[2]])
        print(backend, "Rc", ["%.17g" % float(v) for v in Rc], "r(P0) %.17g" % float(pc.cheby_ball(P[0])[0]))
if len(sys.argv) > 4 and sys.argv[4] == "ulp2":
    i1, i2, i0 = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    res = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[i1], P[i0]); b_ = pc.mldivide(P[i2], P[i0])
        I = pc.intersect(a, b_); D1 = b_.diff(a); D2 = a.diff(b_)
        res[backend] = dict(a=pieces(a), b=pieces(b_), I=pieces(I), D1=pieces(D1), D2=pieces(D2))
    for k in ("a", "b", "I", "D1", "D2"):
        x, y = res["scipy"][k], res["hip"][k]
        print(k, len(x), len(y), [(bool(p.A.shape == q.A.shape and np.array_equal(p.A, q.A)), bool(p.b.shape == q.b.shape and np.array_equal(p.b, q.b))) for p, q in zip(x, y)])
if len(sys.argv) > 4 and sys.argv[4] == "verdicts":
    import hashlib
    import polytope_amd.polytope as pp
    i1, i2, i0 = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    def kp(p): return hashlib.sha1(np.round(np.c_[p.A, p.b], 9).tobytes()).hexdigest()[:6]
    logs = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[i1], P[i0]); b_ = pc.mldivide(P[i2], P[i0])
        log = []
        o_ic, o_cnc = pp.is_convex, pp._clearly_not_convex
        def ic(reg, abs_tol=pp.ABS_TOL):
            out = o_ic(reg, abs_tol); log.append(("+".join(kp(p) for p in reg.list_poly), bool(out[0]), "lp")); return out
        def cnc(group):
            out = o_cnc(group)
            if out: log.append(("+".join(kp(p) for p in group), False, "witness"))
            return out
        pp.is_convex, pp._clearly_not_convex = ic, cnc
        U = pc.union(a, b_, check_convex=True)
        pp.is_convex, pp._clearly_not_convex = o_ic, o_cnc
        logs[backend] = (log, [kp(p) for p in pieces(U)])
    ls, lh = dict((k, (v, w)) for k, v, w in logs["scipy"][0]), dict((k, (v, w)) for k, v, w in logs["hip"][0])
    for k in lh:
        if k in ls and ls[k][0] != lh[k][0]:
            print("verdict differs for group", k, "scipy", ls[k], "hip", lh[k])
    print("groups tested only on scipy:", len([k for k in ls if k not in lh]), "only on hip:", len([k for k in lh if k not in ls]))
    print("pieces scipy", len(logs["scipy"][1]), "hip", len(logs["hip"][1]), "first difference at", next((i for i, (x, y) in enumerate(zip(logs["scipy"][1], logs["hip"][1])) if x != y), None))
if len(sys.argv) > 4 and sys.argv[4] == "eqs":
    import hashlib
    import polytope_amd.polytope as pp
    def kp(p): return hashlib.sha1(np.round(np.c_[p.A, p.b], 9).tobytes()).hexdigest()[:6]
    oeq = pp.Polytope.__eq__
    for backend in ("scipy", "hip", "scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        log = []
        def eq(self, other):
            r = oeq(self, other)
            if self is not other: log.append((kp(self), kp(other), bool(r), round(float(pc.cheby_ball(self)[0]), 6), round(float(pc.cheby_ball(other)[0]), 6)))
            return r
        pp.Polytope.__eq__ = eq
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        out = ops[op](P)
        pp.Polytope.__eq__ = oeq
        print(backend, "pieces", len(pieces(out)), "== evaluated", len(log), "True:", [l for l in log if l[2]])
if len(sys.argv) > 4 and sys.argv[4] == "members":
    i1, i2, i0 = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    res = {}
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
        a = pc.mldivide(P[i1], P[i0]); b_ = pc.mldivide(P[i2], P[i0])
        res[backend] = [pieces(pc.mldivide(pc.Region([m.copy()]), b_)) for m in pieces(a)]
    for k, (x, y) in enumerate(zip(res["scipy"], res["hip"])):
        eq = len(x) == len(y) and all(p.A.shape == q.A.shape and np.array_equal(p.A, q.A) and np.array_equal(p.b, q.b) for p, q in zip(x, y))
        close = len(x) == len(y) and all(p.A.shape == q.A.shape and np.allclose(p.A, q.A, atol=1e-9) and np.allclose(p.b, q.b, atol=1e-9) for p, q in zip(x, y))
        print("member", k, "pieces", len(x), len(y), "bits equal", eq, "1e-9 equal", close)
if len(sys.argv) > 4 and sys.argv[4] == "member2":
    import polytope_amd.polytope as pp
    solvers.default_solver = "scipy"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[1], P[0]); b_ = pc.mldivide(P[2], P[0])
    m = pieces(a)[2]; sub = pieces(b_)[4]
    cut = pieces(pc.mldivide(m.copy(), sub))
    print("cut pieces", len(cut))
    for backend in ("scipy", "hip"):
        solvers.default_solver = backend
        pp._hull_memo.clear(); pp._convex_memo.clear()
        o_cnc = pp._clearly_not_convex
        wit = []
        def cnc(group):
            out = o_cnc(group)
            if out:
                # is the reference's test of the same group really False?
                solvers.default_solver = "scipy"
                truth = bool(pp.is_convex(pc.Region([g_.copy() for g_ in group]))[0])
                solvers.default_solver = backend
                wit.append(truth)
            return out
        pp._clearly_not_convex = cnc
        Pm = pc.Region()
        for x in cut:
            Pm = pc.union(Pm, pp._renormalised(x.copy(), "P"), check_convex=True)
        pp._clearly_not_convex = o_cnc
        print(backend, "after one merge pass:", len(pieces(Pm)), "witness said not convex", len(wit), "times; of those the LP test says convex:", sum(wit))
if len(sys.argv) > 4 and sys.argv[4] == "member2b":
    import polytope_amd.polytope as pp
    solvers.default_solver = "hip"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[1], P[0]); b_ = pc.mldivide(P[2], P[0])
    m = pieces(a)[2]; subs = pieces(b_)
    cut = pc.mldivide(m.copy(), subs[4])
    print("cut", type(cut).__name__, len(pieces(cut)))
    out = pp._passed_untouched(cut, 11)
    print("_passed_untouched(cut, 11) ->", len(pieces(out)))
    touch = pp._cross_touch([m], subs, None, None)
    print("touch row", None if touch is None else np.asarray(touch[0], dtype=int))
    full = pc.mldivide(pc.Region([m.copy()]), b_)
    print("mldivide(Region([m]), b_) ->", len(pieces(full)))
if len(sys.argv) > 4 and sys.argv[4] == "member2c":
    import polytope_amd.polytope as pp
    solvers.default_solver = "hip"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[1], P[0]); b_ = pc.mldivide(P[2], P[0])
    m = pieces(a)[2]; subs = pieces(b_)
    cut = pc.mldivide(m.copy(), subs[4])
    print("memo sizes", len(pp._convex_memo), len(pp._hull_memo))
    keys = [pp._content_key(x) for x in pieces(cut)]
    import itertools
    hits = [(i, j, pp._convex_memo.get(frozenset([keys[i], keys[j]]))) for i, j in itertools.combinations(range(len(keys)), 2) if frozenset([keys[i], keys[j]]) in pp._convex_memo]
    print("pairs of the cut's pieces already in the memo:", hits)
    out1 = pp._passed_untouched(pc.Region([x for x in pieces(cut)]), 11)
    print("with the memo as it is ->", len(pieces(out1)))
    pp._convex_memo.clear(); pp._hull_memo.clear()
    out2 = pp._passed_untouched(pc.Region([x for x in pieces(cut)]), 11)
    print("memos cleared ->", len(pieces(out2)))
    # one pass by hand, verbose
    solvers.default_solver = "hip"
    Pm = pc.Region()
    for x in pieces(cut):
        Pm = pc.union(Pm, x, check_convex=True)
        print("   after adding a piece:", len(pieces(Pm)))
if len(sys.argv) > 4 and sys.argv[4] == "member2d":
    import polytope_amd.polytope as pp
    solvers.default_solver = "hip"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[1], P[0]); b_ = pc.mldivide(P[2], P[0])
    m = pieces(a)[2]; subs = pieces(b_)
    cut = pieces(pc.mldivide(m.copy(), subs[4]))
    solvers.default_solver = "scipy"
    cut_s = pieces(pc.mldivide(m.copy(), subs[4]))
    print("same pieces in the same order (1e-9):", [bool(x.A.shape == y.A.shape and np.allclose(x.A, y.A, atol=1e-9) and np.allclose(x.b, y.b, atol=1e-9)) for x, y in zip(cut, cut_s)])
    for k, x in enumerate(cut):
        fresh = pc.Polytope(x.A.copy(), x.b.copy(), normalize=False)
        lo, hi = fresh.bounding_box
        r, xc = pc.cheby_ball(fresh)
        print(k, "cached bbox", None if x.bbox is None else "ok" if np.allclose(x.bbox[0], lo, atol=1e-7) and np.allclose(x.bbox[1], hi, atol=1e-7) else ("WRONG", x.bbox[0].ravel(), lo.ravel(), x.bbox[1].ravel(), hi.ravel()),
              "| cached ball", x._chebR, "fresh", float(r), "| minrep", x.minrep, "fulldim", x.fulldim)
if len(sys.argv) > 4 and sys.argv[4] == "leaf":
    import polytope_amd.polytope as pp
    from oracle import oracle as O
    solvers.default_solver = "hip"
    P = [pc.Polytope(A.copy(), b.copy()) for A, b in data]
    a = pc.mldivide(P[1], P[0]); b_ = pc.mldivide(P[2], P[0])
    m = pieces(a)[2]; subs = pieces(b_)
    o_rm = pp._reduce_many
    caught = []
    def rm(polys, *a_, **k):
        out = o_rm(polys, *a_, **k); caught.append(([(p.A.copy(), p.b.copy()) for p in polys], out)); return out
    pp._reduce_many = rm
    cut = pieces(pc.mldivide(m.copy(), subs[4]))
    pp._reduce_many = o_rm
    np.set_printoptions(precision=6, linewidth=200, suppress=True)
    for polys, outs in caught:
        for (Aq, bq), q in zip(polys, outs):
            solvers.default_solver = "scipy"
            ref = pc.reduce(pc.Polytope(Aq.copy(), bq.copy(), normalize=False))
            solvers.default_solver = "hip"
            o = O.reduce(Aq, bq)
            same = q is not None and q.A.shape == ref.A.shape and np.allclose(q.A, ref.A, atol=1e-9) and np.allclose(q.b, ref.b, atol=1e-9)
            print("leaf rows", Aq.shape[0], "hip kept", None if q is None else q.A.shape[0], "scipy kept", ref.A.shape[0], "oracle mask", bin(int(o["mask"])), "nlp", o["nlp"], "same:", same)
            if not same:
                np.savez("gpurun_out/leaf_%d.npz" % Aq.shape[0], A=Aq, b=bq)
                print(np.c_[Aq, bq])
