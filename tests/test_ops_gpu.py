"""Parity of the individual HIP kernels (through the C ABI) against plain torch fp32/fp64
references of the same op on the same seeded inputs.  f32 mode: tight tolerance (exact-f32
MFMA / VALU); bf16 mode: inputs are rounded to bf16 first and the tolerance covers the bf16
output rounding + accumulation-order differences (stated per test)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ishara_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16)}
TOL = {"f32": dict(rtol=2e-4, atol=2e-4), "bf16": dict(rtol=3e-2, atol=3e-2)}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(a).to("cuda", dtype).contiguous()


def close(got, ref, name, rtol, atol):
    got = got.detach().float().cpu().double()
    ref = ref.detach().cpu().double()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    worst = float((err - bound).max())
    assert worst <= 0, f"{name}: max abs err {float(err.max()):.3e} (ref scale {float(ref.abs().max()):.3e}), exceeds tol by {worst:.3e}"


# NT: 0 A-stationary kernel where it applies (bf16, K 256/512), else LDS-DMA 128x128 transposed-acc tile kernel; 8192 tile kernel only;
# 8 LDS-DMA 64x128; bit0 register-staged; bit1: register-transposing TN
@pytest.mark.parametrize("regstage", [0, 8192, 8, 1, 3])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,act", [(300, 276, 64, 0), (256, 64, 128, 1), (1408, 256, 768, 0), (130, 512, 60, 2), (64, 32, 8, 0), (3000, 768, 256, 0), (1024, 512, 256, 0), (4096, 128, 128, 0), (1408, 128, 512, 1),
                                       (1000, 256, 512, 1), (333, 512, 128, 0), (900, 256, 256, 2), (50, 512, 512, 0)])
def test_dense_fwd_bwd(lib, dt, M, K, N, act, regstage):
    lib.ishara_debug_force_regstage(regstage)
    try:
        _dense_fwd_bwd(lib, dt, M, K, N, act)
    finally:
        lib.ishara_debug_force_regstage(0)


def _dense_fwd_bwd(lib, dt, M, K, N, act):
    code, tdt = DT[dt]
    if dt == "bf16" and K % 8:
        pytest.skip("bf16 activations always have 16-byte rows (K % 8 == 0); the ragged-K stem input is f32")
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g)).to(tdt)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g).to(tdt)
    Wq = W.to(tdt).float()
    xd, Wd, bd, dyd = x.cuda().contiguous(), dev(W), dev(b), dy.cuda().contiguous()
    y = torch.empty(M, N, dtype=tdt, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    _lib.check(lib.ishara_op_dense_fwd(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, act, scp, stream()))
    ref = x.double() @ Wq.double() + b.double()
    ref = [ref, ref * torch.sigmoid(ref), torch.relu(ref)][act]
    close(y, ref, "dense_fwd", **TOL[dt])
    if act != 0:
        return
    dx = torch.empty(M, K, dtype=tdt, device="cuda")
    dW = torch.zeros(K, N, device="cuda")
    db = torch.zeros(N, device="cuda")
    _lib.check(lib.ishara_op_dense_bwd(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(dyd), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, stream()))
    close(dx, dy.double() @ Wq.double().t(), "dense_dx", **TOL[dt])
    wtol = dict(rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * M ** 0.5)
    close(dW, x.double().t() @ dy.double(), "dense_dW", **wtol)
    close(db, dy.double().sum(0), "dense_db", **wtol)


@pytest.mark.parametrize("regstage", [0, 8192, 8])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,act", [(1408, 256, 768, 0), (1000, 256, 256, 0), (333, 512, 128, 0), (2048, 512, 256, 1), (130, 64, 64, 0), (77, 256, 64, 2)])
def test_dense_fwd_residual(lib, dt, M, K, N, act, regstage):
    """y = act(x W + b) + resid through ishara_op_dense_fwd_ex (the residual epilogues of every NT kernel)."""
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(7 * M + K + N)
    x = torch.randn(M, K, generator=g).to(tdt)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(tdt)
    xd, Wd, bd, rd = x.cuda().contiguous(), dev(W), dev(b), r.cuda().contiguous()
    y = torch.empty(M, N, dtype=tdt, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    lib.ishara_debug_force_regstage(regstage)
    try:
        _lib.check(lib.ishara_op_dense_fwd_ex(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), M, K, N, act, scp, stream()))
    finally:
        lib.ishara_debug_force_regstage(0)
    ref = x.double() @ W.to(tdt).double() + b.double()
    ref = [ref, ref * torch.sigmoid(ref), torch.relu(ref)][act] + r.double()
    close(y, ref, "dense_fwd_residual", **TOL[dt])


@pytest.mark.parametrize("M,K,N,act,with_resid", [(49152, 256, 512, 0, False), (49152 + 200, 256, 256, 1, False), (49152, 256, 768, 0, True), (50000, 256, 512, 2, True),
                                                  (49152, 512, 256, 0, True), (49152 + 300, 512, 256, 1, False), (49152 + 130, 512, 512, 0, True)])
def test_dense_chunked_a_stationary_kernel(lib, M, K, N, act, with_resid):
    """The chunked form of the A-stationary GEMM (gemm_as.hip gemm_nt_as_chunk_kernel: K = 256, M >= 49152 rows, one 12-wave workgroup of
    384 rows per CU, two 64 KB weight stages of 128 columns; K = 512: 8 waves, stages of 64 columns, a 256-row and a 128-row pass) — reached
    only at training-size M, so it gets its own case: against the fp64
    reference on sampled rows (the whole product would be 13 GFLOP of CPU fp64) and BIT-IDENTICAL to the per-step kernel on every element
    (same fragments, same MFMA order, same epilogue arithmetic).  Ragged M (a partial last workgroup), all three stage counts (N = 256 / 512 /
    768), bias + activation + residual epilogues."""
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(M + N + act)
    x = torch.randn(M, K, generator=g).to(tdt)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(tdt) if with_resid else None
    xd, Wd, bd = x.cuda().contiguous(), dev(W), dev(b)
    rd = r.cuda().contiguous() if with_resid else None
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    outs = {}
    try:
        for flags in (3, 51):
            lib.ishara_debug_set_as_flags(flags)
            y = torch.empty(M, N, dtype=tdt, device="cuda")
            _lib.check(lib.ishara_op_dense_fwd_ex(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), M, K, N, act, scp, stream()))
            torch.cuda.synchronize()
            outs[flags] = y
    finally:
        lib.ishara_debug_set_as_flags(-1)
    assert torch.equal(outs[3], outs[51]), f"chunked vs per-step kernel: {(outs[3].float() - outs[51].float()).abs().max().item()}"
    rows = torch.cat([torch.arange(0, 700), torch.arange(M // 2 - 200, M // 2 + 200), torch.arange(M - 500, M)])
    ref = x[rows].double() @ W.to(tdt).double() + b.double()
    ref = [ref, ref * torch.sigmoid(ref), torch.relu(ref)][act]
    if with_resid:
        ref = ref + r[rows].double()
    close(outs[51][rows.cuda()], ref, "dense_chunked", **TOL["bf16"])


@pytest.mark.parametrize("M,K,N,act,with_resid", [(32768, 512, 512, 0, False), (32768, 1024, 512, 1, True), (32768 + 256, 512, 1024, 2, True), (34816, 640, 768, 0, True)])
def test_dense_big_tile_kernel(lib, M, K, N, act, with_resid):
    """The 256 x 256 two-operand tile GEMM (gemm_big.hip: K, N >= 512, M >= 32768 rows in whole 256-row tiles — config #4's shapes) against
    the A-stationary / tile kernels it replaces there (ishara_debug_set_nt_big(0)) on every element, and against the fp64 reference on
    sampled rows.  Same MFMA instruction and the same ascending order over K, so the fp32 accumulators agree bit for bit and the outputs
    differ by at most one bf16 rounding of the epilogue arithmetic.  K = 640: an uneven split of K steps between main loop and tail;
    M = 34816: a row-tile count that is not a multiple of 8 (the plain workgroup order)."""
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(M + N + K + act)
    x = torch.randn(M, K, generator=g).to(tdt)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(tdt) if with_resid else None
    xd, Wd, bd = x.cuda().contiguous(), dev(W), dev(b)
    rd = r.cuda().contiguous() if with_resid else None
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    outs = {}
    try:
        for on in (0, 1, 1, 0):
            lib.ishara_debug_set_nt_big(on)
            y = torch.empty(M, N, dtype=tdt, device="cuda")
            _lib.check(lib.ishara_op_dense_fwd_ex(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), M, K, N, act, scp, stream()))
            torch.cuda.synchronize()
            if on in outs:          # each route twice: both are deterministic, so a run-to-run difference names the route that raced
                assert torch.equal(outs[on], y), f"route big={on} is not run-to-run identical: max diff {(outs[on].float() - y.float()).abs().max().item()}"
            outs[on] = y
    finally:
        lib.ishara_debug_set_nt_big(1)
    d = (outs[0].float() - outs[1].float()).abs()
    tol = 2.0 ** -7 * outs[0].float().abs().clamp_min(1.0)          # one bf16 ulp of the larger magnitude
    if not bool((d <= tol).all()):          # say WHICH route is off: fp64 on the rows where they differ
        bad = torch.nonzero((d > tol).any(1)).flatten().cpu()
        refb = x[bad].double() @ W.to(tdt).double() + b.double()
        refb = [refb, refb * torch.sigmoid(refb), torch.relu(refb)][act]
        if with_resid:
            refb = refb + r[bad].double()
        e0 = (outs[0][bad.cuda()].double().cpu() - refb).abs().max().item()
        e1 = (outs[1][bad.cuda()].double().cpu() - refb).abs().max().item()
        raise AssertionError(f"big tile vs the route it replaces: {len(bad)} rows differ (first {bad[:8].tolist()}), max diff {d.max().item()}; "
                             f"max error against fp64 on those rows: replaced route {e0:.4f}, big tile {e1:.4f}")
    rows = torch.cat([torch.arange(0, 600), torch.arange(M // 2 - 200, M // 2 + 200), torch.arange(M - 400, M)])
    ref = x[rows].double() @ W.to(tdt).double() + b.double()
    ref = [ref, ref * torch.sigmoid(ref), torch.relu(ref)][act]
    if with_resid:
        ref = ref + r[rows].double()
    close(outs[1][rows.cuda()], ref, "dense_big", **TOL["bf16"])


@pytest.mark.parametrize("M,K,N", [(32768, 512, 512), (32768, 512, 1024), (65536, 1024, 512), (32768, 768, 512)])
def test_dense_bwd_big_tile_weight_gradient(lib, M, K, N):
    """The 256 x 256 tile form of the weight-gradient GEMM (gemm_big.hip gemm_tn_big_kernel: K, N >= 512 in whole 256-wide tiles, M >= 32768)
    against the 128 x 128 tile kernel it replaces there (ishara_debug_set_nt_big(0) switches both big kernels off) and against fp64:
    dW = x^T dy, db = column sums of dy, accumulated INTO the outputs (they start non-zero).  The two kernels split M differently, so the
    fp32 sums differ by rounding only."""
    code, tdt = DT["bf16"]
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(tdt)
    dy = (torch.randn(M, N, generator=g) * 0.5).to(tdt)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    xd, dyd, Wd = x.cuda().contiguous(), dy.cuda().contiguous(), dev(W)
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    outs = {}
    try:
        for on in (0, 1, 1):
            lib.ishara_debug_set_nt_big(on)
            dW = torch.full((K, N), 0.25, device="cuda"); db = torch.full((N,), -0.5, device="cuda")
            _lib.check(lib.ishara_op_dense_bwd(code, _lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(dyd), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, stream()))
            torch.cuda.synchronize()
            if on in outs:          # fixed summation order: bit-identical run to run
                assert torch.equal(outs[on][0], dW.cpu()) and torch.equal(outs[on][1], db.cpu()), "the 256 x 256 weight-gradient kernel is not run-to-run identical"
            outs[on] = (dW.cpu(), db.cpu())
    finally:
        lib.ishara_debug_set_nt_big(1)
    refW = x.double().t() @ dy.double() + 0.25
    refb = dy.double().sum(0) - 0.5
    for on in (0, 1):
        close(outs[on][0], refW, f"dW big={on}", rtol=2e-4, atol=2e-3 * (M / 32768) ** 0.5)
        close(outs[on][1], refb, f"db big={on}", rtol=2e-4, atol=2e-3 * (M / 32768) ** 0.5)
    close(outs[1][0], outs[0][0].double(), "dW big vs tile", rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,Cc", [(100, 64), (1000, 256), (77, 512)])
def test_layernorm(lib, dt, M, Cc):
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(Cc)
    x = (torch.randn(M, Cc, generator=g) * 2 + 0.5).to(tdt)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    dy = torch.randn(M, Cc, generator=g).to(tdt)
    xd, dyd, gd, btd = x.cuda(), dy.cuda(), dev(gamma), dev(beta)
    y = torch.empty_like(xd)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    _lib.check(lib.ishara_op_layernorm_fwd(code, _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(btd), C.c_float(1e-6), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), M, Cc, stream()))
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(xr, (Cc,), gr, br, 1e-6)
    close(y, ref, "ln_fwd", **TOL[dt])
    ref.backward(dy.double())
    dx = torch.empty_like(xd)
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    _lib.check(lib.ishara_op_layernorm_bwd(code, _lib.ptr(dyd), _lib.ptr(xd), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), M, Cc, stream()))
    close(dx, xr.grad, "ln_dx", **TOL[dt])
    rtol = TOL[dt]["rtol"]
    close(dg, gr.grad, "ln_dgamma", rtol=rtol, atol=TOL[dt]["atol"] * M ** 0.5)
    close(db, br.grad, "ln_dbeta", rtol=rtol, atol=TOL[dt]["atol"] * M ** 0.5)


def _dw_ref(x, w, bias, inop, padl, C_):
    """x [B,T,Cin] double; depthwise conv reference with the fused input op."""
    if inop == 1:
        u = x * torch.sigmoid(x)
    elif inop == 2:
        u = x[..., :C_] * torch.sigmoid(x[..., C_:])
    else:
        u = x
    k = w.shape[0]
    up = F.pad(u.transpose(1, 2), (padl, k - 1 - padl))
    return F.conv1d(up, w.t().unsqueeze(1), bias, groups=C_).transpose(1, 2)


@pytest.mark.parametrize("variant", ["reg", "lds"])   # register-window kernels (K in 3,5,11,15) vs LDS-tiled kernels
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,T,Cc,k,inop,causal", [(3, 176, 128, 11, 1, True), (2, 64, 256, 3, 1, True), (2, 384, 64, 15, 2, False), (2, 40, 8, 5, 0, True),
                                                 (1, 16, 512, 31, 2, False), (2, 48, 512, 5, 1, True), (2, 104, 32, 15, 2, False)])
def test_dwconv(lib, dt, B, T, Cc, k, inop, causal, variant):
    lib.ishara_debug_force_regstage(4 if variant == "lds" else 0)
    try:
        _dwconv(lib, dt, B, T, Cc, k, inop, causal, variant)
    finally:
        lib.ishara_debug_force_regstage(0)


def _dwconv(lib, dt, B, T, Cc, k, inop, causal, variant):
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(T + k)
    Cin = 2 * Cc if inop == 2 else Cc
    x = torch.randn(B, T, Cin, generator=g).to(tdt)
    w = torch.randn(k, Cc, generator=g) / k ** 0.5
    bias = torch.randn(Cc, generator=g) if not causal else None
    dy = torch.randn(B, T, Cc, generator=g).to(tdt)
    padl = k - 1 if causal else (k - 1) // 2
    xd, dyd, wd = x.cuda(), dy.cuda(), dev(w)
    bd = dev(bias) if bias is not None else None
    y = torch.empty(B, T, Cc, dtype=tdt, device="cuda")
    ssum, ssq = torch.zeros(B, Cc, device="cuda"), torch.zeros(B, Cc, device="cuda")
    _lib.check(lib.ishara_op_dwconv_fwd(code, inop, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), _lib.ptr(ssum), _lib.ptr(ssq), B, T, Cc, k, padl, stream()))
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = bias.double().requires_grad_(True) if bias is not None else None
    ref = _dw_ref(xr, wr, br, inop, padl, Cc)
    close(y, ref, "dw_fwd", **TOL[dt])
    close(ssum, ref.sum(1), "dw_ssum", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * T ** 0.5)
    close(ssq, (ref ** 2).sum(1), "dw_ssq", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * T)
    ref.backward(dy.double())
    dx = torch.empty(B, T, Cin, dtype=tdt, device="cuda")
    dw = torch.zeros(k, Cc, device="cuda")
    dbias = torch.zeros(Cc, device="cuda") if bias is not None else None
    scr = torch.empty(int(lib.ishara_op_dwconv_scratch_bytes(Cc, k)), dtype=torch.uint8, device="cuda") if variant == "reg" else None
    _lib.check(lib.ishara_op_dwconv_bwd(code, inop, _lib.ptr(dyd), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dbias), _lib.ptr(scr), B, T, Cc, k, padl, stream()))
    close(dx, xr.grad, "dw_dx", **TOL[dt])
    close(dw, wr.grad, "dw_dw", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * (B * T) ** 0.5)
    if bias is not None:
        close(dbias, br.grad, "dw_dbias", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * (B * T) ** 0.5)


@pytest.mark.parametrize("dt", ["bf16"])
@pytest.mark.parametrize("B,T,Cc,k,inop,causal", [(9, 384, 512, 11, 1, True), (9, 384, 512, 15, 1, True), (10, 384, 256, 15, 2, False), (9, 200, 128, 11, 0, True),
                                                 (12, 72, 128, 15, 2, False), (9, 512, 1024, 11, 1, True)])
def test_dwconv_streaming_forward(lib, dt, B, T, Cc, k, inop, causal):
    """The streaming K = 11 / 15 forward kernel (elementwise.hip dwconv_stream_kernel: 64-row LDS ring per 128 channels, lane-pair 16-byte
    stores, time ranges per sample) — reached only with caller scratch and more than 8 samples, i.e. the model's own call: outputs and the
    per-sample statistics against the fp64 reference, causal and 'same' padding, ragged T (partial last chunk, ranges that do not divide T),
    GLU input (2C channels) and bias; and bit-identical outputs to the tile kernel it replaces wherever both apply."""
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(T + k + Cc)
    Cin = 2 * Cc if inop == 2 else Cc
    x = torch.randn(B, T, Cin, generator=g).to(tdt)
    w = torch.randn(k, Cc, generator=g) / k ** 0.5
    bias = torch.randn(Cc, generator=g) if not causal else None
    padl = k - 1 if causal else (k - 1) // 2
    xd, wd = x.cuda(), dev(w)
    bd = dev(bias) if bias is not None else None
    y = torch.empty(B, T, Cc, dtype=tdt, device="cuda")
    ssum, ssq = torch.zeros(B, Cc, device="cuda"), torch.zeros(B, Cc, device="cuda")
    sc = torch.empty(int(lib.ishara_op_dwconv_fwd_scratch_bytes(B, T, Cc)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    _lib.check(lib.ishara_op_dwconv_fwd_ex(code, inop, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), _lib.ptr(ssum), _lib.ptr(ssq), scp, B, T, Cc, k, padl, stream()))
    ref = _dw_ref(x.double(), w.double(), bias.double() if bias is not None else None, inop, padl, Cc)
    close(y, ref, "dw_stream_fwd", **TOL[dt])
    close(ssum, ref.sum(1), "dw_stream_ssum", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * T ** 0.5)
    close(ssq, (ref ** 2).sum(1), "dw_stream_ssq", rtol=TOL[dt]["rtol"], atol=TOL[dt]["atol"] * T)
    # the tile kernel on the same inputs: the same fp32 sum per output element (taps in the same order, bias last) -> the same bf16 outputs up to
    # an occasional last-bit difference where the compiler contracted a multiply-add in one kernel only
    y2 = torch.empty_like(y)
    lib.ishara_debug_force_regstage(4)
    try:
        _lib.check(lib.ishara_op_dwconv_fwd_ex(code, inop, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y2), _lib.ptr(ssum), _lib.ptr(ssq), scp, B, T, Cc, k, padl, stream()))
    finally:
        lib.ishara_debug_force_regstage(0)
    torch.cuda.synchronize()
    ndiff = int((y != y2).sum().item())
    assert ndiff <= 1e-3 * y.numel() and (y.float() - y2.float()).abs().max().item() <= 2 ** -6 * max(1.0, y2.float().abs().max().item()), (ndiff, y.numel())


def _attn_ref(qkv, B, H, T, dh, scale, mask):
    d = H * dh
    q4 = qkv.view(B, T, H, 3 * dh).permute(0, 2, 1, 3)
    q, k, v = q4[..., :dh], q4[..., dh:2 * dh], q4[..., 2 * dh:]
    a = torch.softmax(q @ k.transpose(-1, -2) * scale, -1)
    if mask is not None:
        a = a * mask
    return (a @ v).permute(0, 2, 1, 3).reshape(B * T, d)


@pytest.mark.parametrize("impl", [0, 1])   # 0 lane-split VALU kernels, 1 MFMA flash kernels (bf16, dh 32/64)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,T,dh,rate", [(2, 8, 176, 8, 0.0), (2, 4, 64, 32, 0.2), (1, 2, 384, 64, 0.1), (3, 8, 16, 16, 0.0),
                                           (2, 3, 176, 32, 0.2), (1, 2, 512, 64, 0.0), (2, 2, 384, 32, 0.0), (2, 4, 200, 32, 0.1), (1, 8, 384, 32, 0.1)])
def test_attention(lib, dt, B, H, T, dh, rate, impl):
    if impl == 1 and (dt != "bf16" or dh not in (32, 64)):
        pytest.skip("MFMA attention: bf16, head dim 32/64")
    from oracle import rng
    code, tdt = DT[dt]
    d = H * dh
    g = torch.Generator().manual_seed(T + dh)
    qkv = torch.randn(B * T, 3 * d, generator=g).to(tdt)
    dout = torch.randn(B * T, d, generator=g).to(tdt)
    scale = d ** -0.5 * 4.0     # sharper softmax than the model's to exercise the max tracking
    seed, site = 1234, 7
    qd, dd = qkv.cuda(), dout.cuda()
    o = torch.empty(B * T, d, dtype=tdt, device="cuda")
    sc = torch.empty(int(lib.ishara_op_attn_scratch_bytes(B, H, T, dh)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    _lib.check(lib.ishara_op_attn_fwd(code, _lib.ptr(qd), _lib.ptr(o), B, H, T, dh, C.c_float(scale), seed, site, C.c_float(rate), impl, scp, stream()))
    mask = None
    if rate > 0:
        mask = torch.from_numpy(rng.scaled_mask_attn(seed, site, B * H * T, T, rate, dtype=np.float64)).view(B, H, T, T)
    qr = qkv.double().requires_grad_(True)
    ref = _attn_ref(qr, B, H, T, dh, scale, mask)
    close(o, ref, "attn_fwd", **TOL[dt])
    ref.backward(dout.double())
    dqkv = torch.empty(B * T, 3 * d, dtype=tdt, device="cuda")
    _lib.check(lib.ishara_op_attn_bwd(code, _lib.ptr(o), _lib.ptr(dd), _lib.ptr(dqkv), B, H, T, dh, C.c_float(scale), seed, site, C.c_float(rate), impl, scp, stream()))
    btol = TOL[dt] if dt == "f32" else dict(rtol=3e-2, atol=6e-2)     # bf16: o, dout and P are rounded before the 5 backward products
    close(dqkv, qr.grad, "attn_dqkv", **btol)


@pytest.mark.parametrize("B,T,L", [(4, 176, 64), (3, 384, 64), (2, 16, 8)])
def test_ctc_and_decode(lib, B, T, L):
    from oracle import ishara_oracle as O
    Cc = 60
    g = np.random.default_rng(T)
    logits = (g.standard_normal((B, T, Cc)) * 2).astype(np.float32)
    y = np.full((B, L), 59, np.int64)
    for b in range(B):
        n = int(g.integers(0 if b == 0 else 1, min(L, (T - 1) // 2) + 1)) if b < B - 1 else min(L, T // 2)
        y[b, :n] = g.integers(0, 59, n)
        if n >= 4:
            y[b, 1] = y[b, 0]          # repeated label needs the blank transition
    lg = torch.from_numpy(logits).double().requires_grad_(True)
    ref = O.ctc_nll(torch.from_numpy(y), lg)
    ref.sum().backward()
    ld, yd = dev(logits), dev(y, torch.int64)
    nll = torch.empty(B, device="cuda")
    dl = torch.empty(B, T, Cc, device="cuda")
    ws = torch.empty(int(lib.ishara_ctc_workspace_bytes(B, T, L)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.ishara_ctc_loss(_lib.ptr(ld), _lib.ptr(yd), B, T, Cc, L, 59, _lib.ptr(nll), _lib.ptr(dl), C.c_float(1.0), _lib.ptr(ws), stream()))
    close(nll, ref, "ctc_nll", rtol=1e-5, atol=1e-3)
    close(dl, lg.grad, "ctc_grad", rtol=1e-3, atol=2e-5)
    # greedy decode, bit exact (integer work), including ties and the dropped final run
    logits[0, :, :] = 0.0                      # all ties -> argmax 0 everywhere -> empty
    logits[1, -3:, :] = 0.0; logits[1, -3:, 5] = 9.0   # final run of 5s is never emitted
    ld = dev(logits)
    idx = torch.empty(B, T, dtype=torch.int32, device="cuda")
    ln = torch.empty(B, dtype=torch.int32, device="cuda")
    _lib.check(lib.ishara_greedy_decode(_lib.ptr(ld), B, T, Cc, 59, _lib.ptr(idx), _lib.ptr(ln), stream()))
    idx, ln = idx.cpu().numpy(), ln.cpu().numpy()
    for b in range(B):
        want = O.decode_phrase(logits[b])
        assert ln[b] == len(want) and np.array_equal(idx[b, :ln[b]], want), f"decode mismatch sample {b}"


def test_dropout_mask_matches_oracle_rng(lib):
    from oracle import rng
    out = torch.empty(37, 53, device="cuda")
    _lib.check(lib.ishara_dropout_mask(99, 5, 37, 53, C.c_float(0.3), _lib.ptr(out), stream()))
    want = rng.scaled_mask(99, 5, 37, 53, 0.3)
    assert np.array_equal(out.cpu().numpy(), want)
    assert abs(float((out == 0).float().mean()) - 0.3) < 0.05
