"""Wave-level iterations against per-group pivots per stage of the fused reduce kernel at C2 (debug build with
-DPLP_STAGE_STATS: build_variants/libplp_hip_stats.so, PLP_LIB points at it)."""
import ctypes, os, sys
os.environ["PLP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_variants", "libplp_hip_stats.so")
os.environ.setdefault("PLP_REDUCE_MIX", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import polytope_amd as pa
from polytope_amd import _lib
from polytope_amd.synth import random_hpolytopes
lib = _lib.load()
fn = lib.plp_debug_stage_stats if hasattr(lib, "plp_debug_stage_stats") else ctypes.CDLL(os.environ["PLP_LIB"]).plp_debug_stage_stats
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
A, b = random_hpolytopes(100000, 16, 3, seed=0, stream=0)
A = torch.as_tensor(A).cuda(); b = torch.as_tensor(b).cuda()
buf = (ctypes.c_ulonglong * 16)()
fn(buf, 1)
res = pa.reduce_batch(A, b)
torch.cuda.synchronize()
fn(buf, 1)
s = list(buf)
names = ["waves", "F1 wave iters", "F1 group pivots", "F3 wave iters", "F3 group pivots", "F3 group LPs", "F2 loop iters",
         "F2 fin/start blocks", "F2 group pivots", "F2 LPs", "F3 wave LPs", "F1 groups"]
for n, v in zip(names, s):
    print("%-22s %d" % (n, v))
W = s[0]
print("per wave: F1 %.2f iters (mean group %.2f)" % (s[1] / W, s[2] / s[11]))
print("          F3 %.2f iters over %.2f LPs = %.2f per LP (mean group %.2f per LP)" % (s[3] / W, s[10] / W, s[3] / max(s[10], 1), s[4] / max(s[5], 1)))
print("          F2 %.2f loop iters, %.2f fin/start blocks; per group %.2f LPs x %.2f pivots = %.2f pivots"
      % (s[6] / W, s[7] / W, s[9] / s[11], s[8] / max(s[9], 1), s[8] / s[11]))
print("nlp mean", float(res["nlp"].double().mean()))
