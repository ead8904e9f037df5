"""GPU: rank > 0 with the real kernels.  Several ranks share the one MI355X of the box (gloo rendezvous, every rank on
cuda:0): the sharded paths of polytope_amd.dist and bench.py's N > 1 code path run their HIP kernels on every rank
and the reassembled global results are compared with the unsharded calls on every rank.  (No scaling figure is taken
from this: the ranks share one GPU.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script_args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_paths_with_hip_kernels_on_every_rank(world):
    p = _torchrun(world, [os.path.join(ROOT, "tests", "dist_gpu_worker.py")])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert "RANK %d OK" % r in p.stdout, p.stdout[-2000:]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(scaling):
    """bench.py --gpus 2 (two ranks, both on cuda:0, gloo): the line's LP count is the sum over both ranks' batches
    (weak) or the one partitioned batch (strong), and --verify-exchange has every rank compare the gathered buffer
    of the last group, slot by slot and rank by rank, with its own recomputation of every rank's results."""
    import torch
    import polytope_amd as pa
    from polytope_amd.synth import random_hpolytopes
    p = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2", "--batches", "2",
                      "--backend", "gloo", "--scaling", scaling, "--verify-exchange"])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["steps"] == 9
    assert line["exchange_verified"] == {"ranks": 2, "ok": True, "slots_checked": line["exchange_verified"]["slots_checked"]}
    assert line["exchange_verified"]["slots_checked"] >= 1

    def nlp_of(seed, stream):
        A, b = random_hpolytopes(100000, 16, 3, seed=seed, stream=stream)
        return int(pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda())["nlp"].sum().item())
    if scaling == "weak":   # every rank its own batches (stream = rank); 9 steps rotate over batches 0, 1
        want = sum((5 * nlp_of(0, r) + 4 * nlp_of(1, r)) / 9 for r in range(2))
        assert line["config"]["polytopes_per_gpu"] == 100000
    else:                   # one 100k batch per step, split in two contiguous shards
        want = (5 * nlp_of(0, 0) + 4 * nlp_of(1, 0)) / 9
        assert line["config"]["polytopes_per_gpu"] == 50000
    assert abs(line["config"]["lps_per_step"] - want) < 1e-6
