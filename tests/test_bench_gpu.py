"""bench.py contract: one JSON line with the required keys; the N>1 path is rehearsed with two ranks on one GPU over gloo
(`--backend gloo --share-gpu`) — it once dead-locked because only rank 0 ran the profiled step, which contains the
gradient all-reduce."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9


def test_bench_two_ranks_complete():
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline",
           "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["value"] > 0 and "cpu_baseline" not in j


def test_grad_buckets_cover_the_gradient_in_completion_order():
    import numpy as np
    import torch
    from ishara_amd import get_model
    m = get_model(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, kernel_sizes=[3, 5], num_conv_per_block=2, num_heads=2,
                  input_shape=(64, 20), dtype="bf16", max_batch=4, seed=0)
    b = m.grad_buckets()
    assert 1 <= len(b) <= 4
    hi = m.n_train
    for off, cnt in b:                      # completion order = from the end (head) of the flat gradient to its start (stem)
        assert cnt > 0 and off + cnt == hi
        hi = off
    assert hi == 0
    g = np.random.default_rng(0)
    x = g.standard_normal((4, 64, 20)).astype(np.float32)
    y = np.full((4, 64), 59, np.int64); y[:, :5] = g.integers(0, 59, (4, 5))
    m.enable_grad_buckets()
    m.loss_and_gradients(x, y, seed=1)
    ref = m.grads.clone()
    side = torch.cuda.Stream()
    for i in range(len(b)):
        m.wait_grad_bucket(i, side)         # every bucket event was recorded by the backward pass
    side.synchronize()
    assert torch.equal(ref, m.grads)


def test_overlapped_allreduce_matches_flat_allreduce():
    """Two gloo ranks on one GPU: the bucketed, side-stream all-reduce (ISHARA_OVERLAP_ALLREDUCE=1) must give the same
    training trajectory as the single flat all-reduce (the loss after the timed steps is in the bench line)."""
    losses = []
    for flag in ("0", "1"):
        env = dict(os.environ, ISHARA_OVERLAP_ALLREDUCE=flag)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
               "--backend", "gloo", "--share-gpu"]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        losses.append(_last_json(r.stdout)["config"]["loss"])
    assert losses[0] == losses[1], losses
