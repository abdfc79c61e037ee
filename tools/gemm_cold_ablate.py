#!/usr/bin/env python3
"""Phase ablation of the A-stationary NT GEMM on COLD operands (rotation over NB buffer sets, as tools/gemm_cold.py): which phase costs what when
neither the A rows nor the C rows are in the Infinity Cache.  dbg bits (EpiArgs.dbg, compiled into the DBG instantiation only): 1 no epilogue
(no stores), 2 no MFMA, 4 no LDS fragment reads, 8 no weight DMA."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
NB = 12
def timeit(fn, n=24):
    for i in range(NB): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
V = {"full": 0, "no-epilogue": 1, "no-mfma": 2, "no-lds-no-mfma": 6, "only-epilogue(+A load)": 14, "only-A-load": 15, "no-dma": 8}
for (M, K, N) in [(98304, 256, 512), (98304, 512, 256)]:
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NB)]
    ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    # the op's own fixed part (slab memset + weight shadows), to subtract: a GEMM on 256 rows
    f0 = lambda i: lib.ishara_op_dense_fwd_ex(1, _lib.ptr(xs[0]), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(ys[0]), 256, K, N, 0, scp, st())
    base = min(timeit(f0) for _ in range(3))
    out = {}
    for name, bits in V.items():
        lib.ishara_debug_force_regstage(bits << 4)
        for rot in (0, 1):
            f = lambda i: lib.ishara_op_dense_fwd_ex(1, _lib.ptr(xs[i % NB if rot else 0]), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(ys[i % NB if rot else 0]), M, K, N, 0, scp, st())
            out[(name, rot)] = min(timeit(f) for _ in range(3)) - base
    lib.ishara_debug_force_regstage(0)
    print(f"M{M} K{K} N{N} (op overhead {base:.1f} us subtracted):")
    for name in V:
        print(f"   {name:28s} warm {out[(name, 0)]:6.1f} us   cold {out[(name, 1)]:6.1f} us", flush=True)
