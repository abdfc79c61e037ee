import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
from oracle import ishara_oracle as O
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T, L, Cc = 3, 384, 64, 60
g = np.random.default_rng(T)
logits = (g.standard_normal((B, T, Cc)) * 2).astype(np.float32)
y = np.full((B, L), 59, np.int64)
for b in range(B):
    n = int(g.integers(0 if b == 0 else 1, min(L, (T - 1) // 2) + 1)) if b < B - 1 else min(L, T // 2)
    y[b, :n] = g.integers(0, 59, n)
    if n >= 4: y[b, 1] = y[b, 0]
print("lens", [(y[b] != 59).sum() for b in range(B)])
lg = torch.from_numpy(logits).double().requires_grad_(True)
ref = O.ctc_nll(torch.from_numpy(y), lg); ref.sum().backward()
ld, yd = torch.from_numpy(logits).cuda(), torch.from_numpy(y).cuda()
nll = torch.empty(B, device="cuda"); dl = torch.empty(B, T, Cc, device="cuda")
ws = torch.zeros(int(lib.ishara_ctc_workspace_bytes(B, T, L)), dtype=torch.uint8, device="cuda")
_lib.check(lib.ishara_ctc_loss(_lib.ptr(ld), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), _lib.ptr(dl), C.c_float(1.0), _lib.ptr(ws), st()))
torch.cuda.synchronize()
print("nll", nll.cpu().numpy(), "ref", ref.detach().numpy())
d = dl.cpu()
for b in range(B):
    bad = torch.isnan(d[b]).any(1).nonzero().flatten().tolist()
    print("sample", b, "nan frames", bad[:10], len(bad), "max err", float((d[b] - lg.grad[b].float()).abs().nan_to_num(0).max()))
