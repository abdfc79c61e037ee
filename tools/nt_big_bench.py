#!/usr/bin/env python3
"""Timings of the forward dense GEMM at config #4's shapes, A-stationary kernel (big=0) vs the 256 x 256 tile kernel (big=1): tools/nt_big_bench.py [M]"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
SHAPES = [(1024, 512), (512, 1024), (512, 512), (512, 2048), (2048, 512)] if len(sys.argv) < 3 else [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]]
for (K, N) in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    res = {}
    for flags in (0, 1, 3):
        lib.ishara_debug_set_nt_big(flags)
        run = lambda: lib.ishara_op_dense_fwd_ex(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(y), M, K, N, 0, scp, st())
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res[flags] = e0.elapsed_time(e1) / 5 * 1e3
    lib.ishara_debug_set_nt_big(1)
    fl = 2.0 * M * K * N
    print(f"dense M{M} K{K} N{N}: " + "  ".join(f"big={k}: {v:.0f} us ({fl / v / 1e6:.0f} TFLOP/s)" for k, v in res.items()), flush=True)
