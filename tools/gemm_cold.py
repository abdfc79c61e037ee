#!/usr/bin/env python3
"""Warm vs cold timing of the A-stationary NT GEMM and the transposed-read wgrad: the same launch over ONE buffer set (its 150 MB working
set stays in the 256 MB Infinity Cache between launches) against a rotation over NB distinct buffer sets (every launch reads and writes
memory the chip has not touched for NB-1 launches — what a launch sees inside the model)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
NB = 12
def timeit(fn, n):
    for i in range(min(n, NB)): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(98304, 256, 512), (98304, 512, 256), (98304, 256, 256), (98304, 256, 768)]:
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NB)]
    ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    rs = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(NB)]
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    out = {}
    for name, resid, act in (("plain", False, 0), ("+resid", True, 0), ("+swish", False, 1)):
        f = lambda i, rot: lib.ishara_op_dense_fwd_ex(1, _lib.ptr(xs[i % NB if rot else 0]), _lib.ptr(W), _lib.ptr(b), _lib.ptr(rs[i % NB if rot else 0]) if resid else None,
                                                      _lib.ptr(ys[i % NB if rot else 0]), M, K, N, act, scp, st())
        w = min(timeit(lambda i: f(i, False), 24) for _ in range(3))
        c = min(timeit(lambda i: f(i, True), 24) for _ in range(3))
        out[name] = (w, c)
    print(f"NT  M{M} K{K} N{N}: " + "  ".join(f"{k}: warm {w:.1f} cold {c:.1f} us" for k, (w, c) in out.items()), flush=True)
    dys = ys
    dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
    g = lambda i, rot: lib.ishara_op_dense_bwd(1, _lib.ptr(xs[i % NB if rot else 0]), _lib.ptr(W), _lib.ptr(dys[i % NB if rot else 0]), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
    w = min(timeit(lambda i: g(i, False), 24) for _ in range(3))
    c = min(timeit(lambda i: g(i, True), 24) for _ in range(3))
    print(f"TN  M{M} K{K} N{N}: warm {w:.1f} cold {c:.1f} us (wgrad + slab sums + shadow build)", flush=True)
    del xs, ys, rs, dys
