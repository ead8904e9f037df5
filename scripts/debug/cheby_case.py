"""One polytope (npz with A, b) through the stand-alone Chebyshev / bounding-box batches, verifier on and off, next to the oracle.
Usage: gpurun -- 'python scripts/debug/cheby_case.py case.npz'"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import polytope_amd as pa
    from oracle import oracle as O
    O.build()
    z = np.load(sys.argv[1])
    A, b = z["A"], z["b"]
    m, d = A.shape
    np.set_printoptions(precision=9, linewidth=200)
    if os.environ.get("PLP_VERIFY", "1") != "0":
        print("oracle cheby", O.cheby(A, b)[:2], " bbox", O.bounding_box(A, b)[:3])
    dev = torch.device("cuda:0")
    for B in (1, 300):
        At = torch.as_tensor(np.repeat(A[None], B, 0)).to(dev)
        bt = torch.as_tensor(np.repeat(b[None], B, 0)).to(dev)
        ch = pa.cheby_ball_batch(At, bt)
        torch.cuda.synchronize()
        print("verify", os.environ.get("PLP_VERIFY", "1"), "B", B, "cheby status", set(ch["status"].cpu().numpy().tolist()), "r", set(ch["r"].cpu().numpy().tolist()),
              "careful", pa.verify_careful_lps() if hasattr(pa, "verify_careful_lps") else None)
        if B == 1:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "cheby_case_v%s.npz" % os.environ.get("PLP_VERIFY", "1")),
                     xc=ch["xc"].cpu().numpy(), r=ch["r"].cpu().numpy(), status=ch["status"].cpu().numpy())
        bb = pa.bbox_batch(At, bt)
        if bb is not None:
            torch.cuda.synchronize()
            print("   bbox status", set(bb["status"].cpu().numpy().tolist()), "lb", bb["lb"][0].cpu().numpy(), "ub", bb["ub"][0].cpu().numpy())


if __name__ == "__main__":
    main()
    if os.environ.get("PLP_VERIFY", "1") != "0":
        subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, PLP_VERIFY="0"))
