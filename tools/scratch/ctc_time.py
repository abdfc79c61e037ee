import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T, L, Cc = 256, 384, 64, 60
g = np.random.default_rng(1)
logits = torch.from_numpy(g.standard_normal((B, T, Cc)).astype(np.float32)).cuda()
y = np.full((B, L), 59, np.int64)
for b in range(B):
    n = int(g.integers(8, 32)); y[b, :n] = g.integers(0, 59, n)
yd = torch.from_numpy(y).cuda()
nll = torch.empty(B, device="cuda"); dl = torch.empty(B, T, Cc, device="cuda")
ws = torch.zeros(int(lib.ishara_ctc_workspace_bytes(B, T, L)), dtype=torch.uint8, device="cuda")
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("with grad   :", timeit(lambda: lib.ishara_ctc_loss(_lib.ptr(logits), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), _lib.ptr(dl), C.c_float(1.0), _lib.ptr(ws), st())), "us")
print("without grad:", timeit(lambda: lib.ishara_ctc_loss(_lib.ptr(logits), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), None, C.c_float(1.0), _lib.ptr(ws), st())), "us")
