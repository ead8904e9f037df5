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


def _rot_mean(nlp, steps):
    """mean LP count per step of `steps` steps rotating over the batches from batch 0"""
    return sum(nlp[k % len(nlp)] for k in range(steps)) / steps


def _nlp_of(seed, stream):
    import torch
    import polytope_amd as pa
    from polytope_amd.synth import random_hpolytopes
    A, b = random_hpolytopes(100000, 16, 3, seed=seed, stream=stream)
    return int(pa.reduce_batch(torch.as_tensor(A).cuda(), torch.as_tensor(b).cuda())["nlp"].sum().item())


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the environment (the form the driver
    uses): bench.py starts its two ranks itself (both on cuda:0, gloo, since the box has one GPU).  One line, from rank
    0: weak figure as `value`, the partitioned batch as the object `strong`, both exchanges verified slot by slot, every
    polytope of both ranks' batches checked against the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20",
           "--warmup", "5", "--verify-exchange", "--batches", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 20 and line["warmup"] == 5
    cfg = line["config"]
    assert cfg["rccl_ranks_seen"] == 2 and cfg["polytopes_per_gpu"] == 100000 and cfg["regions"] >= 5
    assert line["exchange_verified"]["ok"] is True and line["exchange_verified"]["ranks"] == 2
    assert line["exchange_verified"]["strong"]["ok"] is True
    assert line["parity_ok"] is True and line["parity_checked"] == 2 * 2 * 100000
    S = cfg["timed_steps_per_region"]
    assert S % 20 == 0 and min(cfg["region_ms"]) >= 0.8 * 50.0
    want = sum(_rot_mean([_nlp_of(0, r), _nlp_of(1, r)], S) for r in range(2))
    assert abs(cfg["lps_per_step"] - want) < 1e-6
    st = line["strong"]
    assert st["strong_floor"]["shard_polytopes"] == 50000 and st["strong_floor"]["shard_kernel_ms_alone"] > 0
    assert abs(st["lps_per_step"] - _rot_mean([_nlp_of(0, 0), _nlp_of(1, 0)], st["timed_steps_per_region"])) < 1e-6
    assert st["value"] > 0 and st["ms_per_step"] > 0


def test_bench_two_ranks_under_a_launcher_strong_only():
    """The launcher route (torch.distributed.run, as the contract describes it) with --scaling strong: `value` is the
    partitioned batch, the reassembled global batch is compared in batch order."""
    p = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2", "--batches", "2",
                      "--backend", "gloo", "--scaling", "strong", "--verify-exchange", "--no-parity"])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 9 and "strong" not in line
    assert line["exchange_verified"]["ok"] is True and line["exchange_verified"]["slots_checked"] >= 1
    assert line["config"]["polytopes_per_gpu"] == 50000 and line["config"]["strong_floor"]["shard_polytopes"] == 50000
    S = line["config"]["timed_steps_per_region"]
    assert abs(line["config"]["lps_per_step"] - _rot_mean([_nlp_of(0, 0), _nlp_of(1, 0)], S)) < 1e-6
