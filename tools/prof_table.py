#!/usr/bin/env python3
"""Per-kernel-family table of one profiled training step of the bench workload (HIP-event profiler of the library):
launches, total ms, average us, algorithmic GB/s and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ishara_amd import get_model
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2:      # debug switches of ishara_debug_force_regstage (e.g. 16384: 256 wgrad workgroups)
    from ishara_amd import _lib
    _lib.load().ishara_debug_force_regstage(int(sys.argv[2]))
KW, T_, F_ = bench.MODEL_KW, 384, 224
if len(sys.argv) > 3 and sys.argv[3] == "cfg4":          # BASELINE config #4: d512, 6+6, T512 (not the bench workload; untuned)
    KW, T_, F_ = dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
                      dropout_rate=0.2, num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(512, 224)), 512, 224
model = get_model(**KW, dtype="bf16", max_batch=B, device="cuda:0", seed=0)
g = np.random.default_rng(1)
x = torch.from_numpy(g.standard_normal((B, T_, F_)).astype(np.float32)).cuda()
y = np.full((B, 64), 59, np.int64)
for b in range(B):
    n = int(g.integers(8, 32)); y[b, :n] = g.integers(0, 59, n)
y = torch.from_numpy(y).cuda()
for _ in range(3):
    model.train_on_batch(x, y)
prof = model.profile_step(x, y)
tot = sum(v["ms"] for v in prof.values())
print(f"{'kernel':44s} {'n':>4s} {'ms':>8s} {'avg us':>8s} {'GB/s':>8s} {'TF/s':>7s}")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    avg = v["ms"] / v["launches"] * 1e3
    gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] else 0
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0
    print(f"{k:44s} {v['launches']:4d} {v['ms']:8.3f} {avg:8.1f} {gbs:8.0f} {tf:7.1f}")
print(f"{'total':44s} {'':4s} {tot:8.3f}")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    model.train_on_batch(x, y)
torch.cuda.synchronize(); print(f"wall ms/step (10 steps): {(time.perf_counter() - t0) * 100:.2f}")
