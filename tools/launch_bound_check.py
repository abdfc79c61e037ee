#!/usr/bin/env python3
"""How launch-bound is a configuration?  Wall time per training step vs the sum of its kernels' HIP-event durations (Model.profile_step):
kernel sum << wall means the host's launch rate limits the step (the case a captured training step would help)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import get_model

def data(B, T, F, seed=1):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.standard_normal((B, T, F)).astype(np.float32)).cuda()
    y = np.full((B, 64), 59, np.int64)
    for b in range(B):
        n = int(g.integers(8, 32)); y[b, :n] = g.integers(0, 59, n)
    return x, torch.from_numpy(y).cuda()

CASES = [("notebook 4+4 d256 T176 F276 B64", dict(dim=256, num_conv_squeeze_blocks=4, num_conv_conform_blocks=4, input_shape=(176, 276)), 64),
         ("cfg#1 d64 1+1 T176 F276 B8", dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276)), 8),
         ("cfg#2 d256 2+2 T384 F224 B256", dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, input_shape=(384, 224)), 256)]
for name, kw, B in CASES:
    T, F = kw["input_shape"]
    x, y = data(B, T, F)
    m = get_model(**kw, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
    for _ in range(3): m.train_on_batch(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.train_on_batch(x, y)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
    prof = m.profile_step(x, y)
    ksum = sum(v["ms"] for v in prof.values()); nl = sum(v["launches"] for v in prof.values())
    print(f"{name:36s} wall {wall:7.2f} ms/step   kernel sum {ksum:7.2f} ms   launches {nl}   ({B * T / wall / 1e3:.2f} M frames/s)")
