"""GPU: BASELINE config 4 through the PUBLIC call (own tie order) against the reference's 234 pieces (tests/golden/g12) as SETS:
seeded sample of P's bounding box, membership in the union of either decomposition through the containment kernel."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import polytope_amd.polytope as pc
from polytope_amd import solvers
solvers.default_solver = "hip"
g = np.load(os.path.join(ROOT, "tests", "golden", "g12_config4.npz"))
shape = tuple(int(v) for v in g["c4_shape"])
cells = [pc.box2poly([[idx[k] / shape[k], (idx[k] + 1) / shape[k]] for k in range(4)]) for idx in itertools.product(*[range(n) for n in shape])]
P = pc.Polytope(g["c4_PA"], g["c4_Pb"], normalize=False)
sub = pc.Region(cells[: int(g["c4_nsub"])])
D2 = pc.region_diff(P.copy(), sub)
ref = []
for k in range(int(g["c4_diff_n"])):
    m = int(g["c4_diff_m"][k]); Ab = g["c4_diff_Ab"][k, :m * 5].reshape(m, 5)
    ref.append(pc.Polytope(Ab[:, :4].copy(), Ab[:, 4].copy(), normalize=False))
Dref = pc.Region(ref)
rng = np.random.default_rng(4)
l, u = P.bounding_box
N = 2000000
X = l + rng.random((4, N)) * (u - l)
in2 = D2.contains(X, abs_tol=0) if isinstance(D2, pc.Region) else D2.contains(X, abs_tol=0)
inr = Dref.contains(X, abs_tol=0)
inP = P.contains(X, abs_tol=0)
vol_box = float(np.prod(u - l))
print("pieces", len(D2), "ref", len(ref), "P frac", inP.mean(), "D2 frac", in2.mean(), "ref frac", inr.mean(),
      "D2\\ref", (in2 & ~inr).mean() * vol_box, "ref\\D2", (inr & ~in2).mean() * vol_box, "box volume", vol_box)
insub = sub.contains(X, abs_tol=0)
truth = inP & ~insub
print("truth volume", truth.mean() * vol_box, "| D2: missing", (truth & ~in2).mean() * vol_box, "extra", (in2 & ~truth).mean() * vol_box,
      "| reference: missing", (truth & ~inr).mean() * vol_box, "extra", (inr & ~truth).mean() * vol_box)
