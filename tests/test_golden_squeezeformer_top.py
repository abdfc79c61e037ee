"""CPU: the oracle restatement of the reference's TOP-LEVEL torch model — `Squeezeformer.forward`, squeezeformer/model.py:437-450
(encoder -> bias-free `fc` -> log_softmax) — against vectors produced by running that file itself
(oracle/gen_golden_squeezeformer_top.py -> tests/golden/squeezeformer_top.npz)."""
import os

import numpy as np
import torch

from oracle import squeezeformer_torch_oracle as SO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "squeezeformer_top.npz"))
CFG = {str(k): int(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])}
ECFG = dict(CFG, num_layers=CFG["num_encoder_layers"], half_step_residual=bool(CFG["half_step_residual"]))


def _params(grad=False):
    P = {k[3:]: torch.from_numpy(G[k]).double() for k in G.files if k.startswith("sd/")}
    if grad:
        for k, v in P.items():
            if not k.endswith(("running_mean", "running_var")): v.requires_grad_(True)
    return P


def test_state_dict_is_encoder_plus_fc():
    keys = [k[3:] for k in G.files if k.startswith("sd/")]
    assert keys[-1] == "fc.weight" and G["sd/fc.weight"].shape == (CFG["num_classes"], CFG["encoder_dim"])
    assert all(k.startswith("encoder.") for k in keys[:-1])
    assert [k[len("encoder."):] for k in keys[:-1]] == list(SO.param_shapes(ECFG))


def test_eval_log_probs_match_reference():
    with torch.no_grad():
        y = SO.squeezeformer_top(torch.from_numpy(G["x"]).double(), _params(), ECFG).numpy()
    assert y.shape == G["eval_y"].shape
    np.testing.assert_allclose(y, G["eval_y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.exp(y).sum(-1), 1.0, atol=1e-9)
    lens = G["lengths"]
    want = ((lens >> 2) - 1 >> 1) - 1
    assert np.array_equal(want * 2, G["eval_len"])          # `>> 2 - 1`, `>> 1 - 1`, `* 2`


def test_training_pass_matches_reference_autograd():
    P = _params(grad=True)
    x = torch.from_numpy(G["x"]).double().requires_grad_(True)
    y = SO.squeezeformer_top(x, P, ECFG, training=True, stats={})
    (y * torch.from_numpy(G["train_G"]).double()).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), G["train_y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(x.grad.numpy(), G["train_dx"], rtol=0, atol=2e-4 * np.abs(G["train_dx"]).max() + 1e-7)
    for k in G.files:
        if k.startswith("train_grad/"):
            want = G[k]
            got = P[k[len("train_grad/"):]].grad.numpy()
            assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-6, k
    # a CTC loss on the log-probabilities, as a caller of this model would train it
    with torch.no_grad():
        yc = SO.squeezeformer_top(torch.from_numpy(G["x"]).double(), {k: v.detach() for k, v in P.items()}, ECFG, training=True, stats={})
    loss = torch.nn.functional.ctc_loss(yc.transpose(0, 1), torch.from_numpy(G["ctc_targets"]), torch.from_numpy(G["eval_len"]), torch.tensor([5, 4]),
                                        blank=0, reduction="sum")
    assert abs(float(loss) - float(G["ctc_loss"])) <= 1e-3
