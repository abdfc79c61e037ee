"""GPU: the HIP kernels (through the C-ABI operator entry points) against golden vectors produced by RUNNING the
reference's conformer/conformer.py (tests/golden/conformer_r5.npz) — SURVEY §8a row R5.  The reference modules are
composed from the library's kernels exactly as the reference file composes torch ops (conformer.py:6-73):
  FeedForwardModule    = dense(SiLU) -> dense(+residual) -> LayerNorm
  MultiHeadSelfAttention = packed qkv dense (in_proj re-packed head-major) -> flash attention (scale dh^-0.5)
                           -> out_proj dense(+residual) -> LayerNorm
  ConvolutionModule    = dense -> depthwise conv with fused GLU input op, pad k//2, bias -> BatchNorm (eval: folded
                           into the next dense's weights on the host) -> dense(+residual) -> LayerNorm
Tolerance: f32 kernels <= 1e-4 max-abs (outputs are O(1) after LayerNorm); bf16 <= 0.08."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from ishara_amd import _lib

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "conformer_r5.npz"))
DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Ops:
    def __init__(self, lib, dt):
        self.lib, (self.code, self.tdt) = lib, DT[dt]
        self.keep = []

    def dev(self, a):
        t = torch.as_tensor(np.ascontiguousarray(a)).to("cuda", torch.float32).contiguous()
        self.keep.append(t)
        return t

    def dense(self, x, w_torch, b, act=0, resid=None):
        """x [M,K] (storage dtype), torch Linear weight [N,K] -> Keras layout [K,N]."""
        M, K = x.shape
        Wk = self.dev(np.asarray(w_torch).T)
        N = Wk.shape[1]
        bd = self.dev(b)
        y = torch.empty(M, N, dtype=self.tdt, device="cuda")
        sc = torch.empty(int(self.lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
        scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
        _lib.check(self.lib.ishara_op_dense_fwd_ex(self.code, _lib.ptr(x), _lib.ptr(Wk), _lib.ptr(bd), _lib.ptr(resid), _lib.ptr(y),
                                                   M, K, N, act, scp, _st()), "dense")
        self.keep += [sc, y]
        return y

    def ln(self, x, g, b, eps=1e-5):
        M, Cc = x.shape
        y = torch.empty_like(x)
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        gd, bd = self.dev(g), self.dev(b)
        _lib.check(self.lib.ishara_op_layernorm_fwd(self.code, _lib.ptr(x), _lib.ptr(gd), _lib.ptr(bd), C.c_float(eps), _lib.ptr(y),
                                                    _lib.ptr(mean), _lib.ptr(rstd), M, Cc, _st()), "ln")
        self.keep += [mean, rstd, y]
        return y


def _sd(prefix):
    return {k[3 + len(prefix):]: G[k] for k in G.files if k.startswith("sd/" + prefix)}


def _block(ops, x, sd, B, T, d, heads, ksize):
    lib, code, tdt = ops.lib, ops.code, ops.tdt
    dh = d // heads

    def ffn(x, p):
        h = ops.dense(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], act=1)
        r = ops.dense(h, sd[p + ".linear2.weight"], sd[p + ".linear2.bias"], resid=x)
        return ops.ln(r, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])

    a = ffn(x, "ffn1")
    # attention: re-pack in_proj (q|k|v block-major rows) to the kernels' head-major column order h*3dh + part*dh + i
    Wi, bi = sd["attention.attention.in_proj_weight"], sd["attention.attention.in_proj_bias"]
    perm = np.array([part * d + h * dh + i for h in range(heads) for part in range(3) for i in range(dh)])
    qkv = ops.dense(a, Wi[perm], bi[perm])
    o = torch.empty(B * T, d, dtype=tdt, device="cuda")
    sc = torch.empty(int(lib.ishara_op_attn_scratch_bytes(B, heads, T, dh)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    _lib.check(lib.ishara_op_attn_fwd(code, _lib.ptr(qkv), _lib.ptr(o), B, heads, T, dh, C.c_float(dh ** -0.5), 0, 0, C.c_float(0.0),
                                      1, scp, _st()), "attn")
    r = ops.dense(o, sd["attention.attention.out_proj.weight"], sd["attention.attention.out_proj.bias"], resid=a)
    b = ops.ln(r, sd["attention.layer_norm.weight"], sd["attention.layer_norm.bias"])
    # convolution module
    g2 = ops.dense(b, sd["conv.pointwise_conv1.weight"][:, :, 0], sd["conv.pointwise_conv1.bias"])
    wdw = ops.dev(sd["conv.depthwise_conv.weight"][:, 0, :].T)          # [d,1,k] -> [k,d]
    bdw = ops.dev(sd["conv.depthwise_conv.bias"])
    v = torch.empty(B * T, d, dtype=tdt, device="cuda")
    _lib.check(lib.ishara_op_dwconv_fwd(code, 2, _lib.ptr(g2), _lib.ptr(wdw), _lib.ptr(bdw), _lib.ptr(v), None, None,
                                        B, T, d, ksize, ksize // 2, _st()), "dwconv")
    aa = sd["conv.batch_norm.weight"] / np.sqrt(sd["conv.batch_norm.running_var"] + 1e-5)        # eval BatchNorm = affine
    bb = sd["conv.batch_norm.bias"] - sd["conv.batch_norm.running_mean"] * aa
    W2 = sd["conv.pointwise_conv2.weight"][:, :, 0]                                               # [out,in]
    r = ops.dense(v, W2 * aa[None, :], sd["conv.pointwise_conv2.bias"] + W2 @ bb, resid=b)
    c = ops.ln(r, sd["conv.layer_norm.weight"], sd["conv.layer_norm.bias"])
    e = ffn(c, "ffn2")
    out = ops.ln(e, sd["layer_norm.weight"], sd["layer_norm.bias"])
    ops.keep += [sc, o, v]
    return dict(ffn1=a, attn=b, conv=c, ffn2=e, out=out)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_hip_kernels_reproduce_reference_conformer_outputs(lib, dt):
    d, layers, heads, ksize, exp = [int(v) for v in G["cfg"]]
    B, T, _ = G["x"].shape
    tol = 1e-4 if dt == "f32" else 0.08
    ops = Ops(lib, dt)
    x = torch.from_numpy(G["x"]).reshape(B * T, d).to("cuda", DT[dt][1]).contiguous()
    for i in range(layers):
        taps = _block(ops, x, _sd(f"layers.{i}."), B, T, d, heads, ksize)
        torch.cuda.synchronize()
        if i == 0:
            for name in ("ffn1", "attn", "conv", "ffn2"):
                err = float((taps[name].float().cpu().numpy().reshape(B, T, d) - G[f"blk0_{name}"]).__abs__().max())
                assert err <= tol, f"block 0 {name}: max-abs-err {err:.3e}"
        err = float(np.abs(taps["out"].float().cpu().numpy().reshape(B, T, d) - G[f"blk{i}_out"]).max())
        assert err <= tol, f"block {i} output: max-abs-err {err:.3e}"
        x = taps["out"]
