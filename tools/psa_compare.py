#!/usr/bin/env python3
"""Per-parameter gradient differences between the PSA route and the materialised-operand route (ISHARA_NO_PSA) at B = 64, configs[1] model."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import get_model
KW = dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
          num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(384, 224))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = np.random.default_rng(9)
x = torch.from_numpy(g.standard_normal((B, 384, 224)).astype(np.float32)).cuda()
y = np.full((B, 64), 59, np.int64)
for i in range(B):
    n = int(g.integers(8, 32)); y[i, :n] = g.integers(0, 59, n)
y = torch.from_numpy(y).cuda()
mb = get_model(**KW, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
os.environ["ISHARA_NO_PSA"] = "1"
ma = get_model(**KW, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
del os.environ["ISHARA_NO_PSA"]
ma.set_weights(mb.get_weights())
mb.loss_and_gradients(x, y, seed=13); gb = mb.grads.clone()
ma.loss_and_gradients(x, y, seed=13); ga = ma.grads.clone()
ma.loss_and_gradients(x, y, seed=13); ga2 = ma.grads.clone()
print("A vs A again: identical", bool(torch.equal(ga, ga2)))
rows = []
for name, shape, off, tr in mb.entries:
    if not tr: continue
    n = int(np.prod(shape)); a, b = ga[off:off + n], gb[off:off + n]
    rows.append(((a - b).norm().item() / max(a.norm().item(), 1e-30), name, a.norm().item(), a.abs().max().item()))
rows.sort(reverse=True)
for r in rows[:12]: print(f"{r[0]:.4f}  {r[1]:50s} |g|={r[2]:.3e} max={r[3]:.3e}")
print("flat rel", ((ga - gb).norm() / ga.norm()).item())
