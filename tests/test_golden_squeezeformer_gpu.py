"""GPU: ishara_amd.SqueezeformerEncoder (csrc/squeezeformer_r4.hip through ishara_encoder_forward / _backward)
against vectors produced by RUNNING the reference's squeezeformer/{attention,modules,convolution,encoder}.py
(oracle/gen_golden_squeezeformer.py -> tests/golden/squeezeformer_r4.npz): the eval-mode encoder output and the training-mode
output, input gradient, every parameter gradient and the BatchNorm1d running statistics of the reference's own autograd.
f32: 1e-4 on outputs (O(1) after LayerNorm), 2e-3 of each tensor's max on gradients; bf16: 0.1 / rel-L2 0.12."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "squeezeformer_r4.npz"))
CFG = {str(k): int(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])}


def _encoder(dt):
    from ishara_amd.squeezeformer import SqueezeformerEncoder
    B, T, _ = G["x"].shape
    enc = SqueezeformerEncoder(CFG["input_dim"], CFG["encoder_dim"], CFG["num_layers"], CFG["reduce_layer_index"], CFG["recover_layer_index"],
                               CFG["num_attention_heads"], CFG["feed_forward_expansion_factor"], CFG["conv_expansion_factor"], 0.0, 0.0, 0.0, 0.0,
                               CFG["conv_kernel_size"], bool(CFG["half_step_residual"]), seq_len=T, max_batch=B, dtype=dt)
    enc.load_state_dict({k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd/")})
    return enc


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_squeezeformer_eval_matches_reference(dt, monkeypatch):
    monkeypatch.setenv("ISHARA_WS_GUARD", "1")
    from ishara_amd import _lib
    enc = _encoder(dt).eval()
    y, lens = enc(torch.from_numpy(G["x"]), torch.tensor([90, 83]))
    y = y.cpu().numpy()
    assert y.shape == G["eval_y"].shape and np.array_equal(lens.numpy(), G["eval_len"])
    err = float(np.abs(y - G["eval_y"]).max())
    assert err <= (1e-4 if dt == "f32" else 0.1), f"encoder output max-abs-err {err:.3e}"
    _lib.check(enc._lib.ishara_workspace_guard_check(enc._h), "workspace guard")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_squeezeformer_training_pass_matches_reference_autograd(dt, monkeypatch):
    monkeypatch.setenv("ISHARA_WS_GUARD", "1")
    from ishara_amd import _lib
    enc = _encoder(dt).train()
    x = torch.from_numpy(G["x"]).cuda().requires_grad_(True)
    y = enc(x)
    (y * torch.from_numpy(G["train_G"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    _lib.check(enc._lib.ishara_workspace_guard_check(enc._h), "workspace guard")
    yerr = float(np.abs(y.detach().cpu().numpy() - G["train_y"]).max())
    assert yerr <= (1e-4 if dt == "f32" else 0.1), f"training-mode output max-abs-err {yerr:.3e}"
    dx, want_dx = x.grad.cpu().numpy(), G["train_dx"]
    if dt == "f32":
        assert np.abs(dx - want_dx).max() <= 2e-3 * np.abs(want_dx).max()
    else:
        assert np.linalg.norm(dx - want_dx) <= 0.12 * np.linalg.norm(want_dx)
    grads = enc.grad_state_dict()
    names = [k[len("train_grad/"):] for k in G.files if k.startswith("train_grad/")]
    assert sorted(names) == sorted(grads)
    gscale = max(float(np.abs(G["train_grad/" + n]).max()) for n in names)
    bad = []
    for n in names:
        want, got = G["train_grad/" + n], grads[n].numpy()
        assert got.shape == want.shape, n
        if np.abs(want).max() < 1e-5 * gscale:             # analytically zero (key_proj.bias: a constant added to every score of a row)
            if np.abs(got).max() > (1e-3 if dt == "f32" else 3e-2) * gscale: bad.append((n, float(np.abs(got).max())))
        elif dt == "f32":
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 2e-3: bad.append((n, e))
        else:
            e = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            if e > (0.25 if want.size <= 16 else 0.12): bad.append((n, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]
    after = enc.state_dict()
    for k in G.files:
        if k.startswith("train_stats/"):
            n = k[len("train_stats/"):]
            np.testing.assert_allclose(after[n].numpy(), G[k], rtol=0, atol=(2e-5 if dt == "f32" else 2e-2), err_msg=n)


@pytest.mark.parametrize("case", ["dh32_no_recover", "dh64_full"])
def test_squeezeformer_other_shapes_vs_oracle(case):
    """Head dims 32 / 64, a longer clip, half_step_residual off, an encoder that reduces but never recovers: HIP (f32) against the
    fp64 oracle restatement (itself pinned by the reference-run fixtures) — output, input gradient and parameter gradients."""
    from ishara_amd.squeezeformer import SqueezeformerEncoder
    from oracle import squeezeformer_torch_oracle as SO
    if case == "dh32_no_recover":
        cfg = dict(input_dim=24, encoder_dim=64, num_layers=3, reduce_layer_index=1, recover_layer_index=3, num_attention_heads=2,
                   feed_forward_expansion_factor=2, conv_expansion_factor=2, conv_kernel_size=15, half_step_residual=False)
        B, T = 3, 150
    else:
        cfg = dict(input_dim=16, encoder_dim=128, num_layers=3, reduce_layer_index=1, recover_layer_index=2, num_attention_heads=2,
                   feed_forward_expansion_factor=4, conv_expansion_factor=2, conv_kernel_size=31, half_step_residual=True)
        B, T = 2, 300
    P = SO.init_params(cfg, seed=5, dtype=torch.float64)
    enc = SqueezeformerEncoder(cfg["input_dim"], cfg["encoder_dim"], cfg["num_layers"], cfg["reduce_layer_index"], cfg["recover_layer_index"],
                               cfg["num_attention_heads"], cfg["feed_forward_expansion_factor"], 2, 0.0, 0.0, 0.0, 0.0, cfg["conv_kernel_size"],
                               cfg["half_step_residual"], seq_len=T, max_batch=B, dtype="f32")
    enc.load_state_dict({k: v.float() for k, v in P.items()})
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, cfg["input_dim"], generator=g)
    for k, v in P.items():
        v.requires_grad_(not k.endswith(("running_mean", "running_var")))
    xo = x.double().requires_grad_(True)
    yo, _ = SO.encoder(xo, P, cfg, training=True)
    assert yo.shape[1] == SO.output_length(T, cfg) == enc.T_out
    Gm = torch.randn(yo.shape, generator=g)
    (yo * Gm.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = enc.train()(xg)
    (y * Gm.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert float((y.detach().cpu().double() - yo.detach()).abs().max()) <= 2e-4
    assert float((xg.grad.cpu().double() - xo.grad).abs().max()) <= 2e-3 * float(xo.grad.abs().max())
    grads = enc.grad_state_dict()
    gscale = max(float(v.grad.abs().max()) for k, v in P.items() if v.grad is not None)
    bad = []
    for k, v in P.items():
        if v.grad is None: continue
        want, got = v.grad.numpy(), grads[k].double().numpy()
        if np.abs(want).max() < 1e-6 * gscale:
            if np.abs(got).max() > 1e-3 * gscale: bad.append((k, float(np.abs(got).max())))
        else:
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 3e-3: bad.append((k, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_squeezeformer_reference_default_shapes_vs_oracle(dt):
    """The reference's own default shapes (squeezeformer/encoder.py:30-46: input_dim 80, encoder_dim 512, 8 heads -> head dim 64, ffn x4, conv kernel
    31) on a T = 800 clip (198 frames after the conv2d subsampling), reduce / recover around the middle — with 4 layers instead of 16 so that the
    fp64 oracle's training pass stays in seconds: HIP output, input gradient and every parameter gradient against the oracle restatement (itself
    pinned by the reference-run fixtures), in the parity mode AND in bf16 (the fixtures and the other-shapes test stop at d = 128, f32)."""
    from ishara_amd.squeezeformer import SqueezeformerEncoder
    from oracle import squeezeformer_torch_oracle as SO
    cfg = dict(input_dim=80, encoder_dim=512, num_layers=4, reduce_layer_index=1, recover_layer_index=3, num_attention_heads=8,
               feed_forward_expansion_factor=4, conv_expansion_factor=2, conv_kernel_size=31, half_step_residual=False)
    B, T = 2, 800
    P = SO.init_params(cfg, seed=7, dtype=torch.float64)
    enc = SqueezeformerEncoder(cfg["input_dim"], cfg["encoder_dim"], cfg["num_layers"], cfg["reduce_layer_index"], cfg["recover_layer_index"],
                               cfg["num_attention_heads"], cfg["feed_forward_expansion_factor"], 2, 0.0, 0.0, 0.0, 0.0, cfg["conv_kernel_size"],
                               cfg["half_step_residual"], seq_len=T, max_batch=B, dtype=dt)
    enc.load_state_dict({k: v.float() for k, v in P.items()})
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, cfg["input_dim"], generator=g)
    for k, v in P.items():
        v.requires_grad_(not k.endswith(("running_mean", "running_var")))
    xo = x.double().requires_grad_(True)
    yo, _ = SO.encoder(xo, P, cfg, training=True)
    assert yo.shape[1] == SO.output_length(T, cfg) == enc.T_out == 198
    Gm = torch.randn(yo.shape, generator=g)
    (yo * Gm.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = enc.train()(xg)
    (y * Gm.cuda()).sum().backward()
    torch.cuda.synchronize()
    yerr = float((y.detach().cpu().double() - yo.detach()).abs().max())
    dx, want_dx = xg.grad.cpu().double(), xo.grad
    grads = enc.grad_state_dict()
    gscale = max(float(v.grad.abs().max()) for k, v in P.items() if v.grad is not None)
    bad = []
    if dt == "f32":
        assert yerr <= 5e-4, yerr
        assert float((dx - want_dx).abs().max()) <= 3e-3 * float(want_dx.abs().max())
    else:
        assert yerr <= 0.15, yerr
        assert float((dx - want_dx).norm() / want_dx.norm()) <= 0.15
    for k, v in P.items():
        if v.grad is None: continue
        want, got = v.grad.numpy(), grads[k].double().numpy()
        if np.abs(want).max() < 1e-6 * gscale:
            if np.abs(got).max() > (1e-3 if dt == "f32" else 3e-2) * gscale: bad.append((k, float(np.abs(got).max())))
        elif dt == "f32":
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 5e-3: bad.append((k, e))
        else:
            e = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            if e > (0.3 if want.size <= 16 else 0.15): bad.append((k, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]
