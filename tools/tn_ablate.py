#!/usr/bin/env python3
"""Ablation timings of the wgrad (TN) path only: dW = x^T dy (+ slab sums + shadow build), bench shapes."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(98304, 256, 512), (98304, 512, 256), (98304, 256, 768), (98304, 256, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    res = {}
    V = {"tr": 0, "tr256": 1 << 6, "tr512": 2 << 6, "tr768": 3 << 6, "nomma": 1, "nofrag": 2, "nomma-nofrag": 3, "noload": 4, "noepi": 8, "noloop": 16, "nothing": 7 | 8}
    for rnd in range(3):
        for name, bits in V.items():
            lib.ishara_debug_force_regstage(bits << 8)
            run = lambda: lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    lib.ishara_debug_force_regstage(0)
    print(f"wgrad M{M} K{K} N{N}: " + "  ".join(f"{k}={min(v):.0f}" for k, v in res.items()) + " us")
