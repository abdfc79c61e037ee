#!/usr/bin/env python3
"""Time one dense forward (x[M,K] @ W[K,N] + b) of the HIP library in isolation, with the
ablation bits of EpiArgs.dbg (1 skip epilogue, 2 skip MFMA, 4 skip loads) and the
register-staged vs LDS-DMA kernels, interleaved in one process (guide rule 24)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from ishara_amd import _lib

lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(98304, 256, 512), (98304, 512, 256), (98304, 256, 768), (98304, 256, 256)]
for (M, K, N) in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    r = torch.randn(M, N, device="cuda").bfloat16()
    res = {}
    variants = {"as": 0, "as-noepi": 1 << 4, "as-nomfma": 2 << 4, "as-nolds": 4 << 4, "as-nolds-nomfma": 6 << 4, "as-onlyepi": 14 << 4, "as-onlyloadA": 15 << 4,
                "as-nodma": 8 << 4, "tile128": 8192, "glds64": 8}
    for rnd in range(3):
        for name, flag in variants.items():
            for resid, act in ((None, 0), (r, 0), (None, 1)) if name in ("as", "tile128", "glds64") else ((None, 0),):
                lib.ishara_debug_force_regstage(flag)
                run = lambda: lib.ishara_op_dense_fwd_ex(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(resid) if resid is not None else None, _lib.ptr(y), M, K, N, act, scp, st())
                for _ in range(2):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(name + ("+r" if resid is not None else "") + ("+swish" if act else ""), []).append(e0.elapsed_time(e1) / 10 * 1e3)
    lib.ishara_debug_force_regstage(0)
    gb = (M * K * 2 + M * N * 2) / 1e9
    print(f"M{M} K{K} N{N}: " + "  ".join(f"{k}={min(v):.0f}" for k, v in res.items()) + f" us  [{gb / (min(res['as']) * 1e-6) / 1e3:.2f} TB/s, includes shadow build+memset]")

print("---- TN (wgrad only: dW = x^T dy, + slab reduce + shadow build)")
for (M, K, N) in [(98304, 256, 512), (98304, 512, 256), (98304, 256, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    res = {}
    for rnd in range(3):
        for name, flag in {"tr": 0, "tr-nomma": (1 << 8), "tr-nofrag": (2 << 8), "tr-noload": (4 << 8), "tr-nothing": (7 << 8), "regstage-tn": 2}.items():
            lib.ishara_debug_force_regstage(flag)
            for _ in range(2):
                lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    lib.ishara_debug_force_regstage(0)
    print(f"wgrad M{M} K{K} N{N}: " + "  ".join(f"{k}={min(v):.0f}us" for k, v in res.items()))
