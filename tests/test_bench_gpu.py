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
