#!/usr/bin/env python3
"""Time the attention operators of the HIP library in isolation on the bench shape (B=256, H=8, T=384, dh=32)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, T, dh = 256, 8, 384, 32
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
qkv = (torch.randn(B * T, 3 * H * dh, device="cuda") * 0.5).bfloat16()
o = torch.empty(B * T, H * dh, device="cuda", dtype=torch.bfloat16)
do = torch.randn(B * T, H * dh, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
sc = torch.empty(int(lib.ishara_op_attn_scratch_bytes(B, H, T, dh)) + 256, dtype=torch.uint8, device="cuda")
scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
scale = dh ** -0.5
for rate in (0.0, 0.1):
    f = timeit(lambda: lib.ishara_op_attn_fwd(1, _lib.ptr(qkv), _lib.ptr(o), B, H, T, dh, C.c_float(scale), 7, 3, C.c_float(rate), 1, scp, st()))
    b = timeit(lambda: lib.ishara_op_attn_bwd(1, _lib.ptr(o), _lib.ptr(do), _lib.ptr(dqkv), B, H, T, dh, C.c_float(scale), 7, 3, C.c_float(rate), 1, scp, st()))
    lib.ishara_debug_force_regstage(1 << 16)          # the two-kernel backward (dq, then dk/dv)
    b2 = timeit(lambda: lib.ishara_op_attn_bwd(1, _lib.ptr(o), _lib.ptr(do), _lib.ptr(dqkv), B, H, T, dh, C.c_float(scale), 7, 3, C.c_float(rate), 1, scp, st()))
    lib.ishara_debug_force_regstage(0)
    print(f"rate={rate}: fwd(+qkv split)={f:.0f}us  bwd one-pass={b:.0f}us  bwd two-pass={b2:.0f}us")
