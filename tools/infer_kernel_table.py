#!/usr/bin/env python3
"""Per-kernel-family HIP-event table of ONE eval-mode forward at B=1, T=384 (configs[4]): where the single-clip latency goes."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import get_model, _lib
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
m = get_model(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], input_shape=(384, 276), dtype=dtype, max_batch=1, seed=0)
x = np.random.default_rng(0).standard_normal((1, 384, 276)).astype(np.float32)
for _ in range(3): m(x)
lib = m._lib
lib.ishara_profile_enable(m._h, 1)
m(x)
buf = C.create_string_buffer(1 << 16)
lib.ishara_profile_report(m._h, buf, len(buf))
lib.ishara_profile_enable(m._h, 0)
rows = [l.split() for l in buf.value.decode().splitlines()]
tot = sum(float(r[2]) for r in rows); n = sum(int(r[1]) for r in rows)
print(f"{dtype}: {n} profiled launches, {tot:.3f} ms of kernel time")
for r in sorted(rows, key=lambda r: -float(r[2])): print(f"{r[0]:44s} n={int(r[1]):3d} {float(r[2]) * 1e3:8.1f} us  ({float(r[2]) * 1e3 / int(r[1]):5.1f} us each)")
