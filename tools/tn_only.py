import ctypes as C, sys, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, K, N = 98304, 256, 512
x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
W = torch.randn(K, N, device="cuda") / K ** 0.5
dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
for flag in (0, 7 << 8, 15 << 8, 8 << 8, 31 << 8):
    lib.ishara_debug_force_regstage(flag)
    for _ in range(10):
        lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
    torch.cuda.synchronize()
