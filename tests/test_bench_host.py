"""CPU: the host-side pieces of bench.py that need no GPU -- the full-batch parity check (pool of oracle workers that
regenerate the batches from their seeds) and the self-launcher of `--gpus N`."""
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_parity_check_catches_a_single_wrong_polytope(oracle):
    import bench
    from polytope_amd.synth import random_hpolytopes
    A, b = random_hpolytopes(bench.B_PER_GPU, bench.M_ROWS, bench.DIM, seed=0, stream=3)
    R = oracle.reduce_batch(A, b)
    good = {"keep": R["keep"].view(np.int64).copy(), "flags": R["flags"], "nlp": R["nlp"], "r": R["r"]}
    pool = mp.get_context("fork").Pool(4)
    try:
        out = bench.parity_check(pool, 4, [good], stream=3)
        assert out["ok"] and out["checked"] == bench.B_PER_GPU and out["oracle_lps"] == int(R["nlp"].sum())
        assert out["max_abs_r_err"] == 0.0
        bad = dict(good, keep=good["keep"].copy(), r=good["r"].copy())
        bad["keep"][77777] ^= 2
        bad["r"][5] += 1e-8
        out = bench.parity_check(pool, 4, [bad], stream=3)
        assert not out["ok"] and out["mismatches"] == {"keep": 1, "flags": 0, "nlp": 0, "r": 1}
        # another stream is other data: nearly everything differs
        assert bench.parity_check(pool, 4, [good], stream=4)["mismatches"]["r"] > bench.B_PER_GPU // 2
    finally:
        pool.close()
        pool.join()


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """No GPU here: both ranks must come up with the environment a launcher would give them, refuse to run without a
    device (no CPU fallback) and the launcher must hand their exit code on."""
    import torch
    if torch.cuda.is_available():
        return   # on a GPU box this path is tests/test_dist_gpu.py's
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-parity"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    # (each rank says so; on a loaded host the launcher may stop the slower one before it got to say it)
    n = p.stderr.count("needs a MI355X")
    assert n == 2 or (n == 1 and "stopping the other ranks" in p.stderr), p.stderr[-2000:]
