#!/usr/bin/env python3
"""Training step time of the other BASELINE configurations (bf16, dropout on): config #4 (d512, 6+6, T512) at a few batch sizes,
the notebook's 4+4 model at its B=64 / T=176, config #1.  Same timing bracket as bench.py (device sync on both sides)."""
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

CASES = [
    ("cfg#4 d512 6+6 T512 F224", dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, kernel_sizes=[11, 5, 3], input_shape=(512, 224)), (32, 64, 128)),
    ("notebook 4+4 d256 T176 F276", dict(dim=256, num_conv_squeeze_blocks=4, num_conv_conform_blocks=4, input_shape=(176, 276)), (64, 128)),
    ("cfg#1 d64 1+1 T176 F276", dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276)), (8, 256)),
]
for name, kw, batches in CASES:
    for B in batches:
        try:
            T, F = kw["input_shape"]
            x, y = data(B, T, F)
            m = get_model(**kw, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
            for _ in range(3): m.train_on_batch(x, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10
            for _ in range(n): m.train_on_batch(x, y)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"{name:30s} B={B:4d}: {dt * 1e3:8.2f} ms/step  {B * T / dt / 1e6:6.2f} M frames/s  params {m.n_train:,}  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
            del m; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        except Exception as e:
            print(f"{name:30s} B={B}: FAILED {type(e).__name__}: {e}")
