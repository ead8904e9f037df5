"""find_adjacent_regions' pair LPs at d >= 5: lane-group kernel (PLP_ADJ_WIDE=0, d <= 8) against one pair per wavefront."""
import os, sys, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
def run(A, b, env):
    if env is None: os.environ.pop("PLP_ADJ_WIDE", None)
    else: os.environ["PLP_ADJ_WIDE"] = env
    r = pa.adjacent_pairs(A, b); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): pa.adjacent_pairs(A, b)
    ev[1].record(); torch.cuda.synchronize()
    return r, ev[0].elapsed_time(ev[1]) / 5
for (shape, kind) in [((4, 4, 4, 4, 3), "boxes"), ((4, 4, 3, 3, 2, 2), "boxes"), ((3, 3, 3, 2, 2, 2, 2, 1), "boxes"), ((4, 4, 4, 2, 2, 1, 1, 1, 1, 1), "boxes"),
                      ((600, 20, 6), "random"), ((600, 32, 8), "random"), ((400, 24, 5), "random"), ((60, 12, 6), "random"), ((400, 32, 12), "random")]:
    if kind == "boxes":
        d = len(shape)
        lo = np.array(list(itertools.product(*[range(k) for k in shape])), dtype=float)
        n = lo.shape[0]
        A = np.tile(np.vstack([np.eye(d), -np.eye(d)]), (n, 1, 1)); b = np.concatenate([lo + 1.0, -lo], axis=1)
    else:
        n, m, d = shape
        A, b = random_hpolytopes(n, m, d, seed=5, stream=0)
        b = 0.6 * b + np.einsum("nij,nj->ni", A, 0.35 * np.random.default_rng(1).standard_normal((n, d)))
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    npairs = n * (n - 1) // 2
    r1, t1 = run(At, bt, "1")
    if d <= 8:
        r0, t0 = run(At, bt, "0")
        rd, td = run(At, bt, None)
        print("%s n=%d (%d pairs) rows 2x%d d=%d: lane groups %.3f ms | one pair per wavefront %.3f ms (%.3g pairs/s) | default %.3f | equal %s"
              % (kind, n, npairs, A.shape[1], d, t0, t1, npairs / t1 * 1e3, td, bool(torch.equal(r0, r1))), flush=True)
    else:
        print("%s n=%d (%d pairs) rows 2x%d d=%d: one pair per wavefront %.3f ms (%.3g pairs/s)" % (kind, n, npairs, A.shape[1], d, t1, npairs / t1 * 1e3), flush=True)
