import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, K, N) in [(1408, 256, 768), (1024, 512, 256), (128, 256, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    _lib.check(lib.ishara_op_dense_fwd_ex(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), M, K, N, 0, scp, st()))
    torch.cuda.synchronize()
    ref = x.float() @ W.bfloat16().float() + b + r.float()
    err = (y.float() - ref).abs()
    bad = err > 0.1
    print(M, K, N, "bad frac", bad.float().mean().item())
    cols = bad.any(0).nonzero().flatten().tolist()
    rows = bad.any(1).nonzero().flatten().tolist()
    print(" bad cols", cols[:40], len(cols)); print(" bad rows", rows[:40], len(rows))
    # is the error equal to a residual from elsewhere?
    d = (y.float() - (ref - r.float()))   # what was added as residual
    i = bad.nonzero()[:5].tolist()
    for (m, n) in i:
        print("  at", m, n, "added", d[m, n].item(), "expected", r[m, n].item())
