#!/usr/bin/env python3
"""Which of the two routes is wrong when the big-tile GEMM and the kernels it replaces disagree?  Replays the sequence of tests/test_ops_gpu.py::
test_dense_big_tile_kernel (four shapes, fresh random data every round, so that LDS / cache contents left by the previous launch differ from what a
launch should read) and checks the rows on which the routes differ against fp64."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(32768, 512, 512, 0, False), (32768, 1024, 512, 1, True), (32768 + 256, 512, 1024, 2, True), (34816, 640, 768, 0, True)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for rnd in range(rounds):
    for (M, K, N, act, with_resid) in SHAPES:
        g = torch.Generator().manual_seed(1000 * rnd + M + N + K + act)
        x = torch.randn(M, K, generator=g).bfloat16(); W = torch.randn(K, N, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g).bfloat16() if with_resid else None
        xd, Wd, bd = x.cuda(), W.cuda(), b.cuda()
        rd = r.cuda() if with_resid else None
        sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
        scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
        outs = {}
        for on in (0, 1, 1):
            lib.ishara_debug_set_nt_big(on)
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.ishara_op_dense_fwd_ex(1, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), M, K, N, act, scp, st()))
            torch.cuda.synchronize()
            if on in outs and not torch.equal(outs[on], y.float().cpu()):
                print(f"round {rnd} {M}x{K}x{N}: the big route is not run-to-run identical", flush=True)
            outs[on] = y.float().cpu()
        d = (outs[0] - outs[1]).abs()
        bad = torch.nonzero(d.amax(1) > 0.05 * outs[0].abs().amax(1).clamp_min(1.0)).flatten()
        if len(bad):
            ref = x[bad].double() @ W.bfloat16().double() + b.double()
            ref = [ref, ref * torch.sigmoid(ref), torch.relu(ref)][act]
            if with_resid: ref = ref + r[bad].double()
            e0 = (outs[0][bad].double() - ref).abs().amax(1); e1 = (outs[1][bad].double() - ref).abs().amax(1)
            cols = torch.nonzero(d[bad[0]] > 0.05).flatten()
            print(f"round {rnd} {M}x{K}x{N}: {len(bad)} rows differ (rows {bad[:10].tolist()}, cols of the first {cols[:16].tolist()} .. {len(cols)} cols); max error vs fp64: "
                  f"big=0 {e0.max().item():.3f}, big=1 {e1.max().item():.3f}", flush=True)
        else:
            print(f"round {rnd} {M}x{K}x{N}: routes agree (max diff {d.max().item():.4f})", flush=True)
lib.ishara_debug_set_nt_big(1)
