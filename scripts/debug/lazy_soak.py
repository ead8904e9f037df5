"""Soak of reduce_lazy_kernel against the dense two-rows-per-lane kernel: random shapes d = 9..16, 33..64 rows, ragged
row counts, duplicated / parallel / zero / infeasible rows, box rows (degenerate vertices), unbounded polytopes;
every output bitwise against the one-row twin, against two rows per lane everything but the last bits of xc.  gpurun -- 'python scripts/debug/lazy_soak.py [trials]'"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import polytope_amd as pa
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4242)
bad = 0
nretry = 0
for trial in range(trials):
    d = int(rng.integers(9, 17)); m = int(rng.integers(33, 65)); B = int(rng.integers(50, 400))
    A = rng.standard_normal((B, m, d)); A /= np.linalg.norm(A, axis=2, keepdims=True)
    b = 0.5 + rng.random((B, m))
    kind = trial % 5
    if kind == 1 and m >= 2 * d:      # box rows: degenerate vertices, bounded
        A[:, :2 * d] = np.vstack([np.eye(d), -np.eye(d)])[None]; b[:, :2 * d] = 1.0 + rng.random((B, 1))
    if kind == 2:                     # duplicated and nearly parallel rows
        j = rng.integers(0, m - 1, B); A[np.arange(B), j + 1] = A[np.arange(B), j]; b[np.arange(B), j + 1] = b[np.arange(B), j] + rng.choice([0.0, 1e-9, 0.05], B)
    if kind == 3:                     # zero rows, infeasible rows, tiny / huge scales
        A[::7, 3] = 0.0; b[::7, 3] = rng.choice([1.0, -1.0], len(b[::7, 3])); A[::5] *= 1e-3; b[::5] *= 1e-3; A[1::5] *= 1e3; b[1::5] *= 1e3
    if kind == 4:                     # half-spaces only on one side: unbounded
        A[:, :, 0] = np.abs(A[:, :, 0])
    rows = rng.integers(max(d + 2, m - 12), m + 1, B).astype(np.int32)
    os.environ["PLP_REDUCE_LAZY"] = "0"; ref = pa.reduce_batch(A, b, m=rows)
    os.environ["PLP_REDUCE_R1"] = "1"; one = pa.reduce_batch(A, b, m=rows); os.environ.pop("PLP_REDUCE_R1")
    os.environ["PLP_REDUCE_LAZY"] = "1"; got = pa.reduce_batch(A, b, m=rows)
    for k in ref:
        if not np.array_equal(one[k].view(np.uint8), got[k].view(np.uint8)):
            bad += 1; print("MISMATCH vs one-row twin: trial", trial, "kind", kind, (B, m, d), k, flush=True)
        if k != "xc" and not np.array_equal(ref[k].view(np.uint8), got[k].view(np.uint8)):
            bad += 1; print("MISMATCH vs two rows per lane: trial", trial, "kind", kind, (B, m, d), k, flush=True)
    if not np.allclose(ref["xc"], got["xc"], rtol=0, atol=1e-12, equal_nan=True):
        bad += 1; print("MISMATCH xc beyond 1e-12: trial", trial, flush=True)
    nretry += int((ref["xc"].view(np.uint64) != got["xc"].view(np.uint64)).any(axis=1).sum())
print("lazy soak: %d trials, mismatches %d (centres that differ in the last bits from the two-rows-per-lane kernel: %d)" % (trials, bad, nretry))
sys.exit(1 if bad else 0)
