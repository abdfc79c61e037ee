"""GPU: ishara_amd.Squeezeformer (encoder family in csrc/squeezeformer_r4.hip + the bias-free `fc` and log_softmax through the library's dense
and log-softmax operators) against vectors produced by RUNNING the reference's squeezeformer/model.py:366-450
(oracle/gen_golden_squeezeformer_top.py -> tests/golden/squeezeformer_top.npz): eval-mode log-probabilities and output lengths, the
training-mode log-probabilities, the input gradient and every parameter gradient (fc.weight included) of the reference's autograd, and a
CTC loss evaluated on the product's log-probabilities.  f32: 1e-4 on the log-probabilities, 2e-3 of each tensor's max on gradients;
bf16 encoder: 0.1 / rel-L2 0.12 (the head itself always runs in exact fp32, like the reference's)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "squeezeformer_top.npz"))
CFG = {str(k): int(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])}


def _model(dt):
    from ishara_amd import Squeezeformer
    B, T, _ = G["x"].shape
    m = Squeezeformer(CFG["num_classes"], CFG["input_dim"], CFG["encoder_dim"], CFG["num_encoder_layers"], CFG["reduce_layer_index"], CFG["recover_layer_index"],
                      CFG["num_attention_heads"], CFG["feed_forward_expansion_factor"], CFG["conv_expansion_factor"], 0.0, 0.0, 0.0, 0.0,
                      CFG["conv_kernel_size"], bool(CFG["half_step_residual"]), seq_len=T, max_batch=B, dtype=dt)
    m.load_state_dict({k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd/")})
    return m


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_top_level_eval_matches_reference(dt):
    m = _model(dt).eval()
    y, lens = m(torch.from_numpy(G["x"]), torch.from_numpy(G["lengths"]))
    y = y.cpu().numpy()
    assert y.shape == G["eval_y"].shape and np.array_equal(lens.numpy(), G["eval_len"])
    err = float(np.abs(y - G["eval_y"]).max())
    assert err <= (1e-4 if dt == "f32" else 0.1), f"log-probabilities max-abs-err {err:.3e}"
    assert np.abs(np.exp(y.astype(np.float64)).sum(-1) - 1).max() <= 1e-5
    assert list(m.state_dict()) == [k[3:] for k in G.files if k.startswith("sd/")]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_top_level_training_pass_matches_reference_autograd(dt):
    m = _model(dt).train()
    x = torch.from_numpy(G["x"]).cuda().requires_grad_(True)
    y, lens = m(x, torch.from_numpy(G["lengths"]))
    (y * torch.from_numpy(G["train_G"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    yerr = float(np.abs(y.detach().cpu().numpy() - G["train_y"]).max())
    assert yerr <= (1e-4 if dt == "f32" else 0.1), f"training-mode log-probabilities max-abs-err {yerr:.3e}"
    dx, want_dx = x.grad.cpu().numpy(), G["train_dx"]
    if dt == "f32":
        assert np.abs(dx - want_dx).max() <= 2e-3 * np.abs(want_dx).max()
    else:
        assert np.linalg.norm(dx - want_dx) <= 0.12 * np.linalg.norm(want_dx)
    grads = {"encoder." + k: v for k, v in m.encoder.grad_state_dict().items()}
    grads["fc.weight"] = m.fc_weight.grad.cpu()
    names = [k[len("train_grad/"):] for k in G.files if k.startswith("train_grad/")]
    assert sorted(names) == sorted(grads)
    gscale = max(float(np.abs(G["train_grad/" + n]).max()) for n in names)
    bad = []
    for n in names:
        want, got = G["train_grad/" + n], grads[n].numpy()
        assert got.shape == want.shape, n
        if np.abs(want).max() < 1e-5 * gscale:
            if np.abs(got).max() > (1e-3 if dt == "f32" else 3e-2) * gscale: bad.append((n, float(np.abs(got).max())))
        elif dt == "f32":
            e = float(np.abs(got - want).max() / np.abs(want).max())
            if e > 2e-3: bad.append((n, e))
        else:
            e = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            if e > (0.25 if want.size <= 16 else 0.12): bad.append((n, e))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]
    # the CTC loss a caller trains this model with, evaluated on the product's log-probabilities (torch's ctc_loss as in the generator)
    if dt == "f32":
        with torch.no_grad():
            yc, ylc = m(torch.from_numpy(G["x"]), torch.from_numpy(G["lengths"]))
        loss = torch.nn.functional.ctc_loss(yc.cpu().transpose(0, 1), torch.from_numpy(G["ctc_targets"]), ylc, torch.tensor([5, 4]), blank=0, reduction="sum")
        assert abs(float(loss) - float(G["ctc_loss"])) <= 1e-3 * float(G["ctc_loss"])


def test_top_level_trains_end_to_end():
    """Three SGD steps on `ctc_loss(model(x))` through both autograd functions lower the loss: the head's gradient reaches the encoder,
    the encoder's weight copies follow the optimiser (no manual sync_weights)."""
    m = _model("f32").train()
    x, lens = torch.from_numpy(G["x"]).cuda(), torch.from_numpy(G["lengths"])
    tgt = torch.from_numpy(G["ctc_targets"])
    opt = torch.optim.SGD(m.parameters(), lr=0.02)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        y, yl = m(x, lens)
        loss = torch.nn.functional.ctc_loss(y.transpose(0, 1), tgt.cuda(), yl, torch.tensor([5, 4]), blank=0, reduction="sum")
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
