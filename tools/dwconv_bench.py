#!/usr/bin/env python3
"""Time the depthwise-conv operators of the HIP library in isolation on the bench shape (B=256, T=384, C=512)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T, Cc = 256, 384, 512
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(B * T, Cc, device="cuda").bfloat16()
dy = torch.randn(B * T, Cc, device="cuda").bfloat16()
y = torch.empty_like(x); dx = torch.empty_like(x)
ssum = torch.zeros(B, Cc, device="cuda"); ssq = torch.zeros(B, Cc, device="cuda")
mb = B * T * Cc * 2 / 1e6
for k in (3, 5, 11, 15):
    w = torch.randn(k, Cc, device="cuda")
    dw = torch.zeros(k, Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    sc = torch.empty(int(lib.ishara_op_dwconv_scratch_bytes(Cc, k)) + 256, dtype=torch.uint8, device="cuda")
    res = {}
    for inop in (0, 1):
        res[f"fwd(inop={inop},stats)"] = timeit(lambda: lib.ishara_op_dwconv_fwd(1, inop, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), _lib.ptr(ssum), _lib.ptr(ssq), B, T, Cc, k, k - 1, st()))
        res[f"fwd(inop={inop})"] = timeit(lambda: lib.ishara_op_dwconv_fwd(1, inop, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), None, None, B, T, Cc, k, k - 1, st()))
        res[f"bwd(inop={inop})"] = timeit(lambda: lib.ishara_op_dwconv_bwd(1, inop, _lib.ptr(dy), _lib.ptr(x), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(sc), B, T, Cc, k, k - 1, st()))
    print(f"k={k:2d}: " + "  ".join(f"{n}={v:.0f}us" for n, v in res.items()) + f"   [tensor = {mb:.0f} MB]")
# reference stream: y = swish(x) elementwise through the library's row map is not exposed; use torch copy as the stream yardstick
print(f"torch copy {mb:.0f} MB: {timeit(lambda: y.copy_(x)):.0f} us ;  torch silu: {timeit(lambda: torch.nn.functional.silu(x, inplace=False)):.0f} us")
