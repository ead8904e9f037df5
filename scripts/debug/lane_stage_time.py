"""GPU: stage times of reduce_lane_kernel from builds that stop early (scripts/build_lane_variant.sh with -DPLP_LANE_DBG_*):
us per 100 000 / 25 000 (16,3) polytopes for each library given on the command line."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "in-tree":
            env["PLP_LIB"] = os.path.abspath(lib)
        print("%-28s" % os.path.basename(lib), subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
import torch
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
dev = torch.device("cuda:0")
NB = 6
full = [random_hpolytopes(100000, 16, 3, seed=i) for i in range(NB)]
out = []
for B in (100000, 25000):
    devb = [(torch.as_tensor(A_[:B]).to(dev), torch.as_tensor(b_[:B]).to(dev)) for A_, b_ in full]
    for k in range(5):
        pa.reduce_batch(*devb[k % NB])
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(80):
            pa.reduce_batch(*devb[k % NB])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 80)
    out.append("B=%d %.1f us" % (B, best * 1e3))
print("  ".join(out))
