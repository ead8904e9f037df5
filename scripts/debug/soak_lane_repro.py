"""Regenerate the batch of one trial of scripts/soak_lane.py (same RNG stream) and compare kernels / oracle on it.
   python scripts/debug/soak_lane_repro.py <seed> <trial> [gpu]
   python scripts/debug/soak_lane_repro.py 23 47 golden     writes tests/golden/lane_w4_tilted.npz (inputs only: the nine polytopes of
       that batch on which walk4 left its planes; the expected outputs are the oracle's, computed in the tests)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import soak_lane as SL
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
fams = ["random", "ragged", "unbounded", "dup", "scaled", "flat", "lattice"]
for trial in range(want + 1):
    d = int(rng.choice([1, 2, 3, 3, 3, 4, 4]))
    m = int(rng.integers(d + 1, 33))
    cls = trial % 6
    B = [int(rng.integers(1, 300)), int(rng.integers(2000, 9000)), int(rng.integers(15000, 30000)),
         int(rng.integers(41000, 60000)), int(rng.integers(300, 2000)), int(rng.integers(30001, 36000))][cls]
    if m > 16 or d == 4:
        B = min(B, 36000)
    fam = fams[int(rng.integers(0, len(fams)))]
    force = bool(rng.random() < 0.5)
    gs = None
    if force:
        if rng.random() < 0.4:
            gs = str(rng.choice([4, 8, 16]))
    A, b, mrows = SL.make(rng, B, m, d, fam)
print("trial", want, "d", d, "m", m, "B", B, fam, "force", force, "gs", gs)
if len(sys.argv) > 3 and sys.argv[3] == "golden":
    idx = np.array([947, 6665, 14794, 21054, 21526, 26420, 29568, 32365, 32378])
    assert (seed, want) == (23, 47)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lane_w4_tilted.npz"), A=A[idx], b=b[idx], m=mrows[idx])
    sys.exit(0)
from oracle import oracle as O
O.build()
if len(sys.argv) > 3:
    import torch, polytope_amd as pa
    dev = torch.device("cuda:0")
    def run(env):
        for k_ in ("PLP_REDUCE_LANE", "PLP_REDUCE_LANE_GS"):
            os.environ.pop(k_, None)
        os.environ.update(env)
        rd = pa.reduce_batch(torch.as_tensor(A).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(mrows).to(dev))
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in rd.items()}
    base = run({"PLP_REDUCE_LANE": "1"} if force else {})
    other = run({"PLP_REDUCE_LANE": "0"})
    keep = base["keep"].view(np.uint64); ko = other["keep"].view(np.uint64)
    diff = np.flatnonzero((keep != ko) | (base["nlp"] != other["nlp"]) | (base["flags"] != other["flags"]))
    print("lane kernel vs lane-group kernels: polytopes that differ:", diff[:20], len(diff))
    badk = []
    for k in range(B):
        o = O.reduce(A[k, :mrows[k]], b[k, :mrows[k]])
        if int(o["mask"]) != int(keep[k]) or o["nlp"] != base["nlp"][k]:
            badk.append(k)
            print("poly", k, "lane", hex(int(keep[k])), base["nlp"][k], "group", hex(int(ko[k])), other["nlp"][k], "oracle", hex(int(o["mask"])), o["nlp"], "flags", base["flags"][k], o["flags"])
    np.savez("gpurun_out/soak_repro_%d_%d.npz" % (seed, want), A=A[badk], b=b[badk], m=mrows[badk], idx=np.array(badk))
