"""Dropout in the two torch encoder families (the reference's fixtures are dropout-free, and torch's generator cannot be mirrored):
the HIP path is checked against ITSELF through properties any correct implementation has —
  * a fixed seed reproduces the output bit for bit, another seed changes it, eval mode ignores the rate;
  * the backward pass uses the masks of the forward pass: with the seed fixed, loss(x) = sum(enc(x) * G) is a smooth function whose
    directional derivative along a random v, measured by central differences in f32, equals <dx, v> from ishara_encoder_backward, and the
    same along a random direction in parameter space (every dropout site sits between the parameters and the loss);
    The Squeezeformer front end has ReLUs (kinks under a finite step), so the finite-difference error of the dropout-free twin on the
    same direction is the yardstick there."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(kind, p):
    if kind == "conformer":
        from ishara_amd import ConformerEncoder
        enc = ConformerEncoder(64, num_layers=2, num_heads=4, expansion_factor=2, kernel_size=7, dropout=p, seq_len=40, max_batch=2, dtype="f32", seed=3)
        x = torch.randn(2, 40, 64, generator=torch.Generator().manual_seed(1))
    else:
        from ishara_amd import SqueezeformerEncoder
        enc = SqueezeformerEncoder(16, 32, 3, 1, 2, 4, 2, 2, p, p, p, p, 7, True, seq_len=70, max_batch=2, dtype="f32", seed=3)
        x = torch.randn(2, 70, 16, generator=torch.Generator().manual_seed(1))
    return enc, x.cuda()


@pytest.mark.parametrize("kind", ["conformer", "squeezeformer"])
def test_dropout_is_seeded_and_consistent_between_forward_and_backward(kind):
    enc, x = _make(kind, 0.2)
    enc0, _ = _make(kind, 0.0)
    enc0.load_state_dict(enc.state_dict())
    with torch.no_grad():
        ev = enc.eval()(x).clone()                 # before any training pass moves the BatchNorm running statistics
        ev0 = enc0.eval()(x).clone()
        enc.train()
        sd = enc.state_dict()
        a = enc._forward(x, True, seed=11).clone()
        enc.load_state_dict(sd)                    # same running statistics for the repeat (they do not enter a training-mode output anyway)
        b = enc._forward(x, True, seed=11).clone()
        c = enc._forward(x, True, seed=12).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(ev, ev0), "eval mode must not depend on the dropout rate"
    assert (a - ev).abs().max() > 1e-3, "training mode with rate 0.2 must differ from eval"
    # ---- backward uses the forward's masks: finite differences with the seed fixed
    enc.train()
    g = torch.Generator().manual_seed(5)
    G = torch.randn(a.shape, generator=g).cuda()
    v = torch.randn(x.shape, generator=g).cuda()

    def loss_at(xx):
        with torch.no_grad():
            return float((enc._forward(xx, True, seed=11).double() * G.double()).sum())

    enc._forward(x, True, seed=11)
    dx = enc._backward(G)
    eps = 1e-2
    fd = (loss_at(x + eps * v) - loss_at(x - eps * v)) / (2 * eps)
    an = float((dx.double() * v.double()).sum())
    # control: the same measurement on the dropout-free twin (ReLU kinks and f32 round-off give it a floor of its own)
    enc0.train()

    def loss0_at(xx):
        with torch.no_grad():
            return float((enc0._forward(xx, True, seed=11).double() * G.double()).sum())

    enc0._forward(x, True, seed=11)
    dx0 = enc0._backward(G)
    fd0 = (loss0_at(x + eps * v) - loss0_at(x - eps * v)) / (2 * eps)
    an0 = float((dx0.double() * v.double()).sum())
    floor = abs(fd0 - an0) / max(abs(an0), 1.0)
    assert abs(fd - an) <= (2e-2 + 2 * floor) * max(abs(an), 1.0), (fd, an, fd0, an0)
    # a direction in parameter space
    flat0 = enc.params.clone()
    gp = enc.grads[:enc.n_train].clone()
    w = torch.randn(enc.n_train, generator=torch.Generator().manual_seed(7)).cuda() * flat0[:enc.n_train].abs().mean()

    def loss_w(t):
        with torch.no_grad():
            enc.params[:enc.n_train] = flat0[:enc.n_train] + t * w
            enc.sync_weights()
            return float((enc._forward(x, True, seed=11).double() * G.double()).sum())

    fdw = (loss_w(1e-3) - loss_w(-1e-3)) / 2e-3
    with torch.no_grad():
        enc.params.copy_(flat0); enc.sync_weights()
    anw = float((gp.double() * w.double()).sum())
    assert abs(fdw - anw) <= 3e-2 * max(abs(anw), 1.0), (fdw, anw)
