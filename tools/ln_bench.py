#!/usr/bin/env python3
"""Time the LayerNorm operators of the HIP library in isolation on the bench shape (M=98304, C=256)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, Cc = 98304, 256
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(M, Cc, device="cuda").bfloat16(); dy = torch.randn(M, Cc, device="cuda").bfloat16()
y = torch.empty_like(x); dx = torch.empty_like(x)
ga = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda")
mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
f = timeit(lambda: lib.ishara_op_layernorm_fwd(1, _lib.ptr(x), _lib.ptr(ga), _lib.ptr(be), C.c_float(1e-6), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), M, Cc, st()))
b = timeit(lambda: lib.ishara_op_layernorm_bwd(1, _lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ga), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), M, Cc, st()))
f2 = timeit(lambda: lib.ishara_op_layernorm_fwd(1, _lib.ptr(x), _lib.ptr(ga), _lib.ptr(be), C.c_float(1e-6), _lib.ptr(y), None, None, M, Cc, st()))
print(f"fwd without mean/rstd stores: {f2:.1f} us")
mb = M * Cc * 2 / 1e6
print(f"layernorm fwd {f:.1f} us ({2 * mb / f:.2f} TB/s)   bwd (atomic test path, no scratch) {b:.1f} us   [tensor {mb:.0f} MB]")
print(f"torch copy {timeit(lambda: y.copy_(x)):.1f} us")
