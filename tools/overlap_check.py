#!/usr/bin/env python3
"""Does running the dgrad (NT) and wgrad (TN) GEMMs of one layer on two streams overlap usefully?"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
M = 98304
for (K, N) in [(256, 512), (512, 256), (256, 256)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
    dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    def scratch():
        sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
        return sc, C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    sc1, p1 = scratch(); sc2, p2 = scratch()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h1, h2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
    # dgrad as a forward-shaped NT product: dx[M,K] = dy[M,N] @ Wt[N,K]
    Wt = W.t().contiguous()
    bz = torch.zeros(K, device="cuda")
    dgrad = lambda h: lib.ishara_op_dense_fwd(1, _lib.ptr(dy), _lib.ptr(Wt), _lib.ptr(bz), _lib.ptr(dx), M, N, K, 0, p1, h)
    wgrad = lambda h: lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, p2, h)
    def run(mode, n=10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev = torch.cuda.Event()
        e0.record(s1)
        for _ in range(n):
            if mode == "seq":
                dgrad(h1); wgrad(h1)
            else:
                ev.record(s1); s2.wait_event(ev)
                wgrad(h2); dgrad(h1)
                ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for m in ("seq", "par"): run(m, 3)
    print(f"K{K} N{N}: sequential {run('seq'):.0f} us   two streams {run('par'):.0f} us   (each includes shadow build + memsets)")
