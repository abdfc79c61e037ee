"""CPU: oracle/squeezeformer_torch_oracle.py (rows R1-R4) against vectors produced by RUNNING the reference's
squeezeformer/{attention,modules,convolution,encoder}.py (oracle/gen_golden_squeezeformer.py -> tests/golden/squeezeformer_r4.npz):
eval-mode outputs of the sub-modules and of every layer, and the training-mode output, input gradient, every parameter
gradient and the BatchNorm running statistics from the reference's own autograd."""
import os

import numpy as np
import pytest
import torch

from oracle import squeezeformer_torch_oracle as SO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "squeezeformer_r4.npz"))
CFG = {str(k): int(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])}
CFG["half_step_residual"] = bool(CFG["half_step_residual"])


def _params(dtype=torch.float64, grad=False):
    P = {k[3:]: torch.from_numpy(G[k]).to(dtype) for k in G.files if k.startswith("sd/")}
    if grad:
        for k, v in P.items():
            if not k.endswith(("running_mean", "running_var")): v.requires_grad_(True)
    return P


def test_param_inventory_matches_reference_state_dict():
    shapes = SO.param_shapes(CFG)
    ref = {k[3:]: G[k].shape for k in G.files if k.startswith("sd/")}
    assert list(shapes) == list(ref)
    assert all(tuple(ref[k]) == tuple(shapes[k]) for k in ref)


def test_rel_positional_encoding_table():
    pe = SO.rel_positional_encoding(13, CFG["encoder_dim"]).numpy()
    assert pe.shape == G["pe_T13"].shape == (1, 25, CFG["encoder_dim"])
    np.testing.assert_allclose(pe, G["pe_T13"], rtol=0, atol=1e-6)
    # row r encodes relative position T-1-r: the centre row is position 0 (sin 0 = 0, cos 0 = 1)
    assert np.allclose(pe[0, 12, 0::2], 0) and np.allclose(pe[0, 12, 1::2], 1)


def test_relative_shift_closed_form():
    x = torch.randn(2, 3, 9, 17, dtype=torch.float64)
    assert torch.equal(SO.relative_shift(x), SO.relative_shift_closed_form(x))


def test_eval_forward_matches_reference():
    P = _params()
    x = torch.from_numpy(G["x"]).double()
    with torch.no_grad():
        h = SO.conv2d_subsampling(x, P)
        np.testing.assert_allclose(h.numpy(), G["subsample"], rtol=0, atol=1e-5)
        h = h @ P["input_proj.0.weight"].t() + P["input_proj.0.bias"]
        np.testing.assert_allclose(h.numpy(), G["input_proj"], rtol=0, atol=1e-5)
        s = "layers.0.sequential"
        np.testing.assert_allclose(SO.rel_mhsa(h, P, s + ".0.module", CFG["num_attention_heads"]).numpy(), G["blk0_mhsa"], rtol=0, atol=1e-5)
        b = torch.from_numpy(G["blk0_after_ffn1_ln"]).double()
        np.testing.assert_allclose(SO.conv_module(b, P, s + ".4.module", training=False).numpy(), G["blk0_conv"], rtol=0, atol=1e-5)
        y, taps = SO.encoder(x, P, CFG, training=False)
    for i, t in enumerate(taps):
        np.testing.assert_allclose(t.numpy(), G[f"eval_layer{i}"], rtol=0, atol=2e-5, err_msg=f"layer {i}")
    np.testing.assert_allclose(y.numpy(), G["eval_y"], rtol=0, atol=2e-5)
    assert y.shape[1] == SO.output_length(x.shape[1], CFG)
    # output lengths (convolution.py:68-69, :266-267, encoder.py:162)
    lens = np.array([90, 83])
    assert np.array_equal(((lens >> 2) - 1), G["subsample_len"])
    assert np.array_equal((((lens >> 2) - 1 >> 1) - 1) * 2, G["eval_len"])


def test_training_pass_matches_reference_autograd():
    P = _params(grad=True)
    x = torch.from_numpy(G["x"]).double().requires_grad_(True)
    stats = {}
    y, _ = SO.encoder(x, P, CFG, training=True, stats=stats)
    np.testing.assert_allclose(y.detach().numpy(), G["train_y"], rtol=0, atol=2e-5)
    (y * torch.from_numpy(G["train_G"]).double()).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), G["train_dx"], rtol=0, atol=1e-4 * np.abs(G["train_dx"]).max())
    gscale = max(float(np.abs(G[k]).max()) for k in G.files if k.startswith("train_grad/"))
    for k in G.files:
        if k.startswith("train_grad/"):
            n, want = k[len("train_grad/"):], G[k]
            got = P[n].grad.numpy()
            # key_proj.bias has an analytically zero gradient (a constant added to every score of a row): float32 noise in the reference
            assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-6 * gscale, n
        if k.startswith("train_stats/"):
            n = k[len("train_stats/"):]
            np.testing.assert_allclose(stats[n].numpy(), G[k], rtol=0, atol=1e-5, err_msg=n)
