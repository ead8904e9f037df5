"""A/B of the assignment kernels at C5 (1M points, d = 8): per-call device time for each load form of the few-facets
kernel and for the general kernel.   python scripts/debug/assign_ab.py [F ...]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, json, torch
sys.path.insert(0, ".")
import polytope_amd as pa
from polytope_amd import synth
F = int(sys.argv[1])
N, d = 1000000, 8
X, nrm, off = synth.quickhull_workload(N, d=d, F=F, seed=0)
dev = torch.device("cuda:0")
Xt, nt, ot = (torch.as_tensor(v).to(dev) for v in (X, nrm, off))
for _ in range(5):
    pa.assign_batch(Xt, nt, ot, 1e-7)
torch.cuda.synchronize()
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pa.assign_batch(Xt, nt, ot, 1e-7)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"F": F, "us_per_call_back_to_back": e0.elapsed_time(e1) / reps * 1e3}))
'''

for F in [int(a) for a in sys.argv[1:]] or [9]:
    for env in ({}, {"PLP_ASSIGN_PPT": "1"}, {"PLP_ASSIGN_PPT": "2"}, {"PLP_ASSIGN_PPT": "4"}, {"PLP_ASSIGN_SMALL": "0"}):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD, str(F)], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"env": env, **(json.loads(line[-1]) if line else {"error": out.stderr[-300:]})}), flush=True)
