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


# ---------------------------------------------------------------------------------------------------------------------
# The product path: ishara_amd.ConformerEncoder (csrc/conformer_r5.hip through ishara_encoder_forward / _backward)
# against the same reference-generated vectors — eval-mode outputs per block, and the TRAINING-mode pass of the fixture
# (BatchNorm1d batch statistics on the GPU, nothing folded on the host): output, input gradient and every parameter
# gradient of the reference's own autograd.
# ---------------------------------------------------------------------------------------------------------------------
def _encoder(dt, layers, dropout=0.0):
    from ishara_amd.conformer import ConformerEncoder
    d, _, heads, ksize, exp = [int(v) for v in G["cfg"]]
    B, T, _ = G["x"].shape
    enc = ConformerEncoder(d, layers, heads, exp, ksize, dropout, seq_len=T, max_batch=B, dtype=dt)
    sd = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd/")}
    enc.load_state_dict({k: v for k, v in sd.items() if int(k.split(".")[1]) < layers})
    return enc


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_product_encoder_eval_matches_reference(dt):
    layers = int(G["cfg"][1])
    enc = _encoder(dt, layers).eval()
    y = enc(torch.from_numpy(G["x"])).cpu().numpy()
    err = float(np.abs(y - G[f"blk{layers - 1}_out"]).max())
    assert err <= (1e-4 if dt == "f32" else 0.08), f"encoder output max-abs-err {err:.3e}"
    # eval mode leaves the running statistics untouched
    after = enc.state_dict()
    assert np.array_equal(after["layers.0.conv.batch_norm.running_mean"].numpy(), G["sd/layers.0.conv.batch_norm.running_mean"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_product_encoder_training_pass_matches_reference_autograd(dt):
    enc = _encoder(dt, 1).train()
    x = torch.from_numpy(G["x"]).cuda().requires_grad_(True)
    y = enc(x)
    (y * torch.from_numpy(G["train_G"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    yerr = float(np.abs(y.detach().cpu().numpy() - G["train_y"]).max())
    assert yerr <= (1e-4 if dt == "f32" else 0.08), f"training-mode output max-abs-err {yerr:.3e}"
    dx, want_dx = x.grad.cpu().numpy(), G["train_dx"]
    if dt == "f32":
        assert np.abs(dx - want_dx).max() <= 1e-3 * np.abs(want_dx).max()
    else:
        assert np.linalg.norm(dx - want_dx) <= 0.08 * np.linalg.norm(want_dx)
    grads = enc.grad_state_dict()
    names = [k[len("train_grad/"):] for k in G.files if k.startswith("train_grad/")]
    assert sorted(names) == sorted(grads)
    gscale = max(float(np.abs(G["train_grad/" + n]).max()) for n in names)
    bad = []
    for n in names:
        want, got = G["train_grad/" + n], grads[n].numpy()
        assert got.shape == want.shape, n
        if np.abs(want).max() < 1e-6 * gscale:             # analytically zero (depthwise bias in front of BatchNorm)
            if np.abs(got).max() > (1e-3 if dt == "f32" else 3e-2) * gscale: bad.append((n, float(np.abs(got).max())))
        elif dt == "f32":
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 1e-3: bad.append((n, e))
        else:
            e = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            if e > 0.1: bad.append((n, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:8]
    # flat.grad (what a torch optimiser steps) is the library's gradient buffer
    assert torch.equal(enc.flat.grad, enc.grads[:enc.n_train])
    # BatchNorm1d running statistics: momentum 0.1 on the new value, unbiased variance
    sd_after = enc.state_dict()
    assert not np.array_equal(sd_after["layers.0.conv.batch_norm.running_var"].numpy(), G["sd/layers.0.conv.batch_norm.running_var"])


def test_product_encoder_running_stats_follow_torch_rule():
    """nn.BatchNorm1d (conformer.py:43): running = 0.9 running + 0.1 batch, with the UNBIASED batch variance — checked against
    the oracle restatement's conv module evaluated on the same block input."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import conformer_torch_oracle as RO
    enc = _encoder("f32", 1).train()
    sd = {k[3:]: torch.from_numpy(G[k]).double() for k in G.files if k.startswith("sd/layers.0.")}
    x = torch.from_numpy(G["x"]).double()
    with torch.no_grad():
        a = RO.ffn(x, sd, "layers.0.ffn1"); b = RO.mhsa(a, sd, "layers.0.attention", int(G["cfg"][2]))
        p = "layers.0.conv"
        d = x.shape[-1]
        u = b @ sd[p + ".pointwise_conv1.weight"][:, :, 0].t() + sd[p + ".pointwise_conv1.bias"]
        u = u[..., :d] * torch.sigmoid(u[..., d:])
        w = sd[p + ".depthwise_conv.weight"]
        u = torch.nn.functional.conv1d(u.transpose(1, 2), w, sd[p + ".depthwise_conv.bias"], padding=w.shape[-1] // 2, groups=d).transpose(1, 2)
        mean, var = u.mean(dim=(0, 1)), u.reshape(-1, d).var(dim=0, unbiased=True)
    with torch.no_grad():
        enc(torch.from_numpy(G["x"]))
    after = enc.state_dict()
    want_m = 0.9 * sd[p + ".batch_norm.running_mean"] + 0.1 * mean
    want_v = 0.9 * sd[p + ".batch_norm.running_var"] + 0.1 * var
    np.testing.assert_allclose(after[p + ".batch_norm.running_mean"].numpy(), want_m.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(after[p + ".batch_norm.running_var"].numpy(), want_v.numpy(), rtol=0, atol=2e-5)


def test_forward_after_an_optimizer_step_uses_the_new_weights():
    """`out = enc(x); loss.backward(); opt.step(); enc(x)` — the reference torch module (conformer.py:76-87) reads its parameters at
    every forward.  The library multiplies MFMA-typed COPIES of the flat fp32 parameters: the forward pass must notice that
    torch's version counter of `enc.flat` moved and re-derive them itself, in f32 and in bf16 mode."""
    for dt in ("f32", "bf16"):
        enc = _encoder(dt, 1).train()
        x = torch.from_numpy(G["x"]).cuda()
        opt = torch.optim.SGD(enc.parameters(), lr=0.05)
        y0 = enc(x)
        y0.square().mean().backward()
        opt.step()
        with torch.no_grad():
            enc.eval()
            auto = enc(x).clone()                  # no manual sync_weights()
            enc.sync_weights()
            manual = enc(x).clone()
        assert torch.equal(auto, manual), dt
        # and the step really changed the function (a stale copy would reproduce the old output)
        stale = _encoder(dt, 1).eval()
        with torch.no_grad():
            before = stale(x)
        assert (auto - before).abs().max() > 1e-3, dt


def test_one_live_graph_per_encoder_and_eval_mode_gradients_are_errors():
    """The library keeps the saved activations of its LAST training forward only: the backward of an older graph must raise
    instead of returning gradients computed from the newer pass's activations; an eval-mode forward of an input that requires
    grad must raise instead of returning a constant."""
    from ishara_amd._lib import IsharaError
    enc = _encoder("f32", 1).train()
    x1 = torch.from_numpy(G["x"]).cuda().requires_grad_(True)
    x2 = (torch.from_numpy(G["x"]).cuda() * 0.5).requires_grad_(True)
    y1 = enc(x1)
    y2 = enc(x2)
    with pytest.raises(IsharaError, match="one live autograd graph"):
        (y1.sum() + y2.sum()).backward()
    y3 = enc(x1)                                   # the normal order still works afterwards
    y3.sum().backward()
    assert x1.grad is not None and torch.isfinite(x1.grad).all()
    enc.eval()
    with pytest.raises(IsharaError, match="eval-mode forward is not differentiable"):
        enc(x1)
    with torch.no_grad():
        enc(x1)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_product_encoder_at_reference_scale_vs_oracle(dt):
    """dim 512, 8 heads (head dim 64), ffn x4, depthwise kernel 31, T = 384 — the scale the reference's commented-out training loop builds
    (conformer.py:89-90 `ConformerEncoder(dim=512, num_layers=6, num_heads=8, expansion_factor=4, kernel_size=31, ...)`), 2 layers so that the fp64 oracle's training pass stays in seconds: output, input
    gradient and every parameter gradient against the oracle restatement (pinned by the reference-run fixtures, which are d = 64)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import conformer_torch_oracle as RO
    from ishara_amd.conformer import ConformerEncoder, default_state_dict
    d, layers, heads, ksize, exp, B, T = 512, 2, 8, 31, 4, 2, 384
    enc = ConformerEncoder(d, layers, heads, exp, ksize, 0.0, seq_len=T, max_batch=B, dtype=dt, seed=4)
    g = np.random.default_rng(9)
    sd = {k: v.numpy().copy() for k, v in enc.state_dict().items()}
    for k in sd:                                    # non-trivial norms and biases
        if "norm" in k and k.endswith("weight"): sd[k] = (1.0 + 0.2 * g.standard_normal(sd[k].shape)).astype(np.float32)
        elif k.endswith("bias"): sd[k] = (0.1 * g.standard_normal(sd[k].shape)).astype(np.float32)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    P = {k: torch.from_numpy(v).double().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in sd.items()}
    x = torch.from_numpy(g.standard_normal((B, T, d)).astype(np.float32))
    Gm = torch.from_numpy(g.standard_normal((B, T, d)).astype(np.float32))
    xo = x.double().requires_grad_(True)
    yo, _ = RO.encoder(xo, P, layers, heads, training=True)
    (yo * Gm.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = enc.train()(xg)
    (y * Gm.cuda()).sum().backward()
    torch.cuda.synchronize()
    yerr = float((y.detach().cpu().double() - yo.detach()).abs().max())
    dx, want_dx = xg.grad.cpu().double(), xo.grad
    if dt == "f32":
        assert yerr <= 3e-4, yerr
        assert float((dx - want_dx).abs().max()) <= 2e-3 * float(want_dx.abs().max())
    else:
        assert yerr <= 0.12, yerr
        assert float((dx - want_dx).norm() / want_dx.norm()) <= 0.12
    grads = enc.grad_state_dict()
    gscale = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    bad = []
    for k, v in P.items():
        if v.grad is None: continue
        want, got = v.grad.numpy(), grads[k].double().numpy()
        if np.abs(want).max() < 1e-6 * gscale:
            if np.abs(got).max() > (1e-3 if dt == "f32" else 3e-2) * gscale: bad.append((k, float(np.abs(got).max())))
        elif dt == "f32":
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 3e-3: bad.append((k, e))
        else:
            e = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            if e > 0.12: bad.append((k, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]
