"""find_adjacent_regions on the 1000-cell grid of config 4: where the second call's time goes (cProfile + stage timers)."""
import cProfile, itertools, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polytope_amd as pc
from polytope_amd import prop2partition as p2p, batch
import torch
pc.solvers.default_solver = "hip"
shape = (10, 10, 5, 2)
cells = [pc.box2poly([[i[k] / shape[k], (i[k] + 1) / shape[k]] for k in range(4)]) for i in itertools.product(*[range(n) for n in shape])]
part = pc.Region(cells)
for _ in range(3):
    t0 = time.perf_counter(); adj = p2p.find_adjacent_regions(part); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("call: %.3f ms, nnz %d" % ((t1 - t0) * 1e3, adj.nnz))
pr = cProfile.Profile(); pr.enable() if os.environ.get("PROFILE") else None
for _ in range(20): p2p.find_adjacent_regions(part)
pr.disable() if os.environ.get("PROFILE") else None
pstats.Stats(pr).sort_stats("cumulative").print_stats(18) if os.environ.get("PROFILE") else None
import time
def t(fn, reps=30):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
regions = p2p._regions_of(part)
print("find_adjacent_regions %.3f ms" % t(lambda: p2p.find_adjacent_regions(part)))
print("  _regions_of %.3f  _members_of %.3f  _device_pairs_ok %.3f  _table_of %.3f" % (
    t(lambda: p2p._regions_of(part)), t(lambda: p2p._members_of(regions)), t(lambda: p2p._device_pairs_ok(regions)), t(lambda: pc.polytope._table_of(regions).dev())))
got = p2p._pair_list_device(regions, "adjacent", pc.polytope.ABS_TOL, diagonal=True)
print("  _pair_list_device %.3f  _lil_from_pairs %.3f" % (t(lambda: p2p._pair_list_device(regions, "adjacent", pc.polytope.ABS_TOL, diagonal=True)),
      t(lambda: p2p._lil_from_pairs(len(regions), got[0], got[1], np.int8, diagonal=not got[2]))))
At, bt, mt = pc.polytope._table_of(regions).dev()
def kern():
    M = batch.adjacent_pairs(At, bt, m=mt, abs_tol=pc.polytope.ABS_TOL); torch.cuda.synchronize(); return M
M = kern()
print("  kernel+sync %.3f  nonzero+cpu %.3f" % (t(kern), t(lambda: torch.nonzero(M).cpu().numpy())))
