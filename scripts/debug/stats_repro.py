"""The wrong-result episode of round 3 (an instrumented -DPLP_STAGE_STATS build of the first presolve version
returned garbage verdicts for whole tiles, only at batch sizes that put more than one wavefront on a SIMD).
Runs the C2 batch through a given library (PLP_LIB) in a child process, several times and at several batch sizes, and
compares keep / flags / nlp with the in-tree library's (which is oracle-exact on this batch).
    python scripts/debug/stats_repro.py build_variants/a40_stats.so [build_variants/a40_prod.so ...]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import ctypes, os
from polytope_amd import _lib
_l = ctypes.CDLL(_lib.LIB_PATH)          # an older library lacks the newer entry points: bind only what it has
for _n in list(_lib.SIGNATURES):
    if not hasattr(_l, _n):
        del _lib.SIGNATURES[_n]
import polytope_amd as pa
from polytope_amd.synth import random_hpolytopes
out = {}
for B in (4096, 32768, 65536, 100000):
    A, b = random_hpolytopes(B, 16, 3, seed=0, stream=0)
    At, bt = torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda()
    for rep in range(3):
        res = pa.reduce_batch(At, bt)
        torch.cuda.synchronize()
        out["%d_%d" % (B, rep)] = np.c_[res["keep"].cpu().numpy().view(np.int64), res["flags"].cpu().numpy(), res["nlp"].cpu().numpy()]
np.savez(sys.argv[1], **out)
'''
libs = [None] + sys.argv[1:]
got = {}
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["PLP_LIB"] = os.path.abspath(lib)
    path = "/tmp/stats_repro_%s.npz" % (os.path.basename(lib) if lib else "intree")
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True)
    if r.returncode:
        print(lib, "FAILED", r.stderr[-400:])
        continue
    import numpy as np
    got[lib] = dict(np.load(path))
ref = got[None]
for lib in libs[1:]:
    if lib not in got:
        continue
    for key in sorted(ref, key=lambda s: tuple(int(v) for v in s.split("_"))):
        a, b = ref[key], got[lib][key]
        bad = np.nonzero((a != b).any(1))[0]
        tiles = np.unique(bad // 16)
        print(json.dumps({"lib": os.path.basename(lib), "batch_rep": key, "polytopes_differ": int(bad.size), "tiles": int(tiles.size),
                          "first": bad[:6].tolist()}))
