#!/usr/bin/env python3
"""Phase split of the CTC kernel at the bench shape (B=256, T=384, C=60, L=64 labels of 8-31 symbols): loss only (phases 0-2: row
log-sum-exp, alpha / beta recursions) against loss + gradient (phase 3: posterior scatter + dlogits)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T, Cc, L = 256, 384, 60, 64
g = np.random.default_rng(0)
logits = torch.from_numpy((g.standard_normal((B, T, Cc)) * 2).astype(np.float32)).cuda()
y = np.full((B, L), 59, np.int64)
for b in range(B):
    n = int(g.integers(8, 32)); y[b, :n] = g.integers(0, 59, n)
yd = torch.from_numpy(y).cuda()
nll = torch.empty(B, device="cuda"); dl = torch.empty(B, T, Cc, device="cuda")
ws = torch.empty(int(lib.ishara_ctc_workspace_bytes(B, T, L)), dtype=torch.uint8, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
full = timeit(lambda: lib.ishara_ctc_loss(_lib.ptr(logits), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), _lib.ptr(dl), C.c_float(1.0), _lib.ptr(ws), st()))
loss = timeit(lambda: lib.ishara_ctc_loss(_lib.ptr(logits), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), None, C.c_float(1.0), _lib.ptr(ws), st()))
half = timeit(lambda: lib.ishara_ctc_loss(_lib.ptr(logits[:, :192].contiguous()), _lib.ptr(yd), B, 192, Cc, L, 59, _lib.ptr(nll), None, C.c_float(1.0), _lib.ptr(ws), st()))
print(f"ctc B{B} T{T}: loss+grad {full:.0f} us, loss only {loss:.0f} us (phase 3 = {full - loss:.0f} us), loss only at T=192 {half:.0f} us -> {(loss - half) / 192 * 1e3:.0f} ns per frame of the recursions")
