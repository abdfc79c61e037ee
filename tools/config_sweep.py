#!/usr/bin/env python3
"""Sanity sweep over model shapes outside the benchmark configuration (BASELINE configs #1, #4, odd widths / lengths):
three training steps each, finite decreasing loss, bf16 vs f32 logits agreement."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import get_model

def data(B, T, F, seed=1):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.standard_normal((B, T, F)).astype(np.float32)).cuda()
    y = np.full((B, 64), 59, np.int64)
    for b in range(B):
        n = int(g.integers(4, min(32, T // 3))); y[b, :n] = g.integers(0, 59, n)
    return x, torch.from_numpy(y).cuda()

CASES = [
    ("cfg#1 d64 1+1 T176 F276", dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276)), 8),
    ("cfg#4 d512 6+6 T512 F224", dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, kernel_sizes=[11, 5, 3], input_shape=(512, 224)), 16),
    ("d128 H4 T128", dict(dim=128, num_heads=4, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(128, 276)), 4),
    ("d192 H6 T200 (ragged)", dict(dim=192, num_heads=6, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(200, 224)), 3),
    ("d384 H8 e4 k31 T384", dict(dim=384, num_heads=8, expansion_factor=4, transformer_kernel_size=31, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(384, 224)), 4),
    ("notebook 4+4 d256 T176 B64", dict(dim=256, num_conv_squeeze_blocks=4, num_conv_conform_blocks=4, input_shape=(176, 276)), 64),
]
for name, kw, B in CASES:
    try:
        T, F = kw["input_shape"]
        x, y = data(B, T, F)
        out = {}
        for dt in ("bf16", "f32"):
            m = get_model(**kw, dropout_rate=0.2, dtype=dt, max_batch=B, seed=0)
            m.optimizer.learning_rate = 1e-3
            out[dt] = m(x, training=False).float().cpu()
            t0 = time.perf_counter()
            losses = [float(m.train_on_batch(x, y, seed=5 + i).item()) for i in range(3)]
            torch.cuda.synchronize()
            ok = np.isfinite(losses).all() and losses[-1] < losses[0]
            out[dt + "_loss"] = losses
            del m
        err = float((out["bf16"] - out["f32"]).abs().max())
        print(f"{name:34s} params ok  bf16 loss {['%.2f' % l for l in out['bf16_loss']]}  f32 loss {['%.2f' % l for l in out['f32_loss']]}  bf16-f32 logits err {err:.3f}  {'OK' if ok and err < 0.25 else 'CHECK'}")
    except Exception as e:
        print(f"{name:34s} FAILED: {type(e).__name__}: {e}")
