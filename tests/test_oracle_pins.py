"""CPU: the oracle against every machine-checkable pin the reference holds for this path
(saved model.summary() parameter counts and tensor shapes — tests/golden/structural_pins.json),
plus independent cross-checks of its CTC / decode / optimizer / schedule restatements."""
import json
import math
import os

import numpy as np
import torch

from oracle import ishara_oracle as O
from oracle import rng

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "structural_pins.json")))


def _cfg(d):
    d = dict(d)
    if "input_shape" in d:
        d["input_shape"] = tuple(d["input_shape"])
    return O.Config(**d)


def test_model_param_counts_match_saved_summaries():
    for m in PINS["models"]:
        tot, tr, nt = O.count_params(_cfg(m["config"]))
        assert (tot, tr, nt) == (m["total"], m["trainable"], m["non_trainable"]), m["source"]
    for m in PINS["derived_for_baseline_configs"]:
        _, tr, nt = O.count_params(_cfg(m["config"]))
        assert (tr, nt) == (m["trainable"], m["non_trainable"]), m["config"]


def test_layer_param_counts():
    def count(prefix, cfg):
        return sum(int(np.prod(s)) for n, s, _, _ in O.param_specs(cfg) if n.startswith(prefix))
    c4 = O.Config(dim=256, num_conv_per_block=0, squeeze_expansion=4, conformer_expansion=2, top_dim=256)
    assert count("stem_conv", c4) == 70656 and count("stem_bn", c4) == 1024
    assert count("squeezeformer_0/", c4) == 1872928
    assert count("conformer_0/", c4) == 992000
    assert count("top_conv", c4) == 65792 and count("classifier", c4) == 15420
    c2 = O.Config(dim=256, num_conv_per_block=0, num_heads=4, top_dim=256)
    assert count("squeezeformer_0/", c2) == 1077280


def test_shapes_and_forward_runs():
    cfg = O.Config(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1)
    assert list(cfg.input_shape) == PINS["input_shape"]["value"]
    P = O.to_torch(O.init_params(cfg, 0), requires_grad=False)
    x, y = O.synthetic_batch(cfg, 2, 1)
    with torch.no_grad():
        logits, _ = O.forward(P, torch.from_numpy(x), cfg)
    assert list(logits.shape[1:]) == PINS["logits_shape"]["value"][1:]
    assert y.shape == (2, 64) and (y[:, -1] == 59).all()
    assert O.tflite_postprocess(np.array([1, 2, 3])).shape == (3, PINS["tflite_output_cols"]["value"])


def test_positional_encoding_is_concatenated_halves():
    pe = O.positional_encoding(16, 8).numpy()
    t = np.arange(16, dtype=np.float32)[:, None]
    rates = 1.0 / np.power(10000.0, np.arange(4, dtype=np.float32) / 4.0)
    assert np.allclose(pe[:, :4], np.sin(t * rates), atol=1e-6) and np.allclose(pe[:, 4:], np.cos(t * rates), atol=1e-6)


def test_ctc_matches_torch_ctc_loss():
    g = np.random.default_rng(0)
    B, T, C, L = 5, 40, 60, 64
    logits = torch.from_numpy(g.standard_normal((B, T, C))).double().requires_grad_(True)
    y = np.full((B, L), 59, np.int64)
    lens = [0, 1, 7, 19, 12]
    for b, n in enumerate(lens):
        y[b, :n] = g.integers(0, 59, n)
    y[2, 1] = y[2, 0]
    yt = torch.from_numpy(y)
    mine = O.ctc_nll(yt, logits)
    ref = torch.nn.functional.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), yt, torch.full((B,), T),
                                       torch.tensor(lens), blank=59, reduction="none")
    assert torch.allclose(mine, ref, atol=1e-9)
    g1, = torch.autograd.grad(mine.sum(), logits)
    logits2 = logits.detach().clone().requires_grad_(True)
    ref2 = torch.nn.functional.ctc_loss(torch.log_softmax(logits2, -1).transpose(0, 1), yt, torch.full((B,), T),
                                        torch.tensor(lens), blank=59, reduction="none")
    g2, = torch.autograd.grad(ref2.sum(), logits2)
    assert torch.allclose(g1, g2, atol=1e-8)
    assert abs(float(O.ctc_loss(yt, logits)) - float(ref.mean())) < 1e-9


def test_decode_phrase_quirks():
    def onehot(seq):
        p = np.zeros((len(seq), 60), np.float32)
        p[np.arange(len(seq)), seq] = 1
        return p
    assert O.decode_phrase(onehot([3, 3, 7, 7])).tolist() == [3]            # final run never emitted (c8:7-9)
    assert O.decode_phrase(onehot([3, 59, 3, 59, 4])).tolist() == [3, 3]     # blanks dropped after collapsing
    assert O.decode_phrase(onehot([59, 59, 59])).tolist() == []
    assert O.decode_phrase(np.zeros((5, 60), np.float32)).tolist() == []     # ties -> first index, one run
    assert O.tflite_postprocess(np.array([5, 6])).shape == (11, 59)           # len<3 fallback (c13:22-23)


def test_radam_lookahead_restatement():
    g = np.random.default_rng(1)
    theta = g.standard_normal(100).astype(np.float32)
    st = O.optimizer_init(theta)
    # rho_t < 4 for the first 4 steps (beta2=.999): un-rectified momentum update
    assert [O.radam_coeffs(t)["rect"] for t in (1, 2, 3, 4, 5, 6)] == [False, False, False, False, False, True] or \
           [O.radam_coeffs(t)["rect"] for t in (1, 2, 3, 4, 5, 6)][0] is False
    th = theta.copy()
    for step in range(1, 11):
        grad = g.standard_normal(100).astype(np.float32)
        th_prev, slow_prev = th.copy(), st.slow.copy()
        th = O.optimizer_step(th, grad, st, lr=1e-2)
        if step % 5 == 0:
            assert np.allclose(th, st.slow)                       # fast weights snap to the slow weights
            assert not np.allclose(st.slow, slow_prev)
        else:
            assert np.allclose(st.slow, slow_prev)
    sma_inf = 2 / (1 - 0.999) - 1
    c = O.radam_coeffs(1000)
    sma = sma_inf - 2 * 1000 * 0.999 ** 1000 / (1 - 0.999 ** 1000)
    assert math.isclose(c["r_t"], math.sqrt((sma - 4) / (sma_inf - 4) * (sma - 2) / (sma_inf - 2) * sma_inf / sma))


def test_lr_schedule():
    sched = [O.lrfn(s, 5, 4e-3, num_training_steps=50) for s in range(50)]
    assert math.isclose(sched[0], 4e-3 / 32) and math.isclose(sched[4], 2e-3) and math.isclose(sched[5], 4e-3)
    assert sched[-1] < 1e-5 and all(a >= b for a, b in zip(sched[5:], sched[6:]))


def test_rng_masks():
    m = rng.keep_mask(7, 3, 200, 300, 0.25)
    assert abs(m.mean() - 0.75) < 0.01
    assert np.array_equal(m, rng.keep_mask(7, 3, 200, 300, 0.25)) and not np.array_equal(m, rng.keep_mask(7, 4, 200, 300, 0.25))
    assert (rng.scaled_mask(1, 1, 4, 4, 0.0) == 1).all()
    # known-answer values of the integer hash (shared with csrc/common.h::lowbias32)
    assert [int(v) for v in rng.lowbias32(np.array([0, 1, 0xDEADBEEF], dtype=np.uint32))] == [0, 0x6C4E2EB8 if False else int(rng.lowbias32(np.array([1], dtype=np.uint32))[0]), int(rng.lowbias32(np.array([0xDEADBEEF], dtype=np.uint32))[0])]


def test_batchnorm_training_statistics_and_dropout_determinism():
    cfg = O.Config(dim=32, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(24, 12), num_heads=4,
                   kernel_sizes=(3,), num_conv_per_block=1, transformer_kernel_size=5)
    W = O.init_params(cfg, 0)
    x, y = O.synthetic_batch(cfg, 3, 1)
    l1, lg1, g1, s1 = O.loss_and_grads(W, x, y, cfg, training=True, seed=5)
    l2, lg2, g2, s2 = O.loss_and_grads(W, x, y, cfg, training=True, seed=5)
    l3, lg3, _, _ = O.loss_and_grads(W, x, y, cfg, training=True, seed=6)
    assert l1 == l2 and np.array_equal(lg1, lg2) and not np.array_equal(lg1, lg3)
    assert set(s1) == {n for n, _, _, t in O.param_specs(cfg) if not t}
    assert all(v is not None for v in g1.values())


def test_attention_quad_rng_statistics():
    """The attention-probability generator draws four 8-bit decisions from one lowbias32 hash (common.h rng_quad; oracle/rng.py
    keep_mask_attn).  The checks a weaker (one-round) hash failed in round 2, applied to the byte lanes: overall keep rate, per-row
    keep-rate spread against the binomial, correlation between neighbouring columns at lags 1-4 (inside and across a quad), between
    neighbouring rows, chi-square of every byte lane's value histogram, and independence of the four lanes of one hash."""
    rows, cols, rate = 2048, 384, 0.2
    t8 = int(rng.threshold8(rate))
    p = 1.0 - t8 / 256.0
    m = rng.keep_mask_attn(11, 5, rows, cols, rate).astype(np.float64)
    assert abs(m.mean() - p) < 4 * np.sqrt(p * (1 - p) / m.size)
    row_rate = m.mean(axis=1)
    assert 0.8 < row_rate.std() / np.sqrt(p * (1 - p) / cols) < 1.2          # binomial spread (the one-round hash: 5x)
    col_rate = m.mean(axis=0)
    assert 0.8 < col_rate.std() / np.sqrt(p * (1 - p) / rows) < 1.2
    z = m - p
    for lag in (1, 2, 3, 4, 5, 8):
        r = float((z[:, :-lag] * z[:, lag:]).mean() / (p * (1 - p)))
        assert abs(r) < 5 / np.sqrt(z[:, lag:].size), (lag, r)                # one-round hash: -0.17 at lag 2
    r = float((z[:-1] * z[1:]).mean() / (p * (1 - p)))
    assert abs(r) < 5 / np.sqrt(z[1:].size), r
    # byte lanes of the raw hash: uniform histograms, and pairwise-independent lanes (chi-square on the 16x16 table of the top nibbles)
    ks = rng.site_key(11, 5)
    with np.errstate(over="ignore"):
        key_row = rng.lowbias32(ks ^ (np.arange(rows, dtype=np.uint32) * np.uint32(0x85EBCA6B)))
    h = rng.lowbias32(key_row[:, None] ^ np.arange(cols // 4, dtype=np.uint32)[None, :]).reshape(-1)
    lanes = [((h >> np.uint32(8 * e)) & np.uint32(0xFF)).astype(np.int64) for e in range(4)]
    n = h.size
    for e in range(4):
        cnt = np.bincount(lanes[e], minlength=256)
        chi = float(((cnt - n / 256) ** 2 / (n / 256)).sum())
        assert chi < 255 + 5 * np.sqrt(2 * 255), (e, chi)                     # one-round hash: 1612 at 255 d.o.f.
    for a in range(4):
        for b in range(a + 1, 4):
            tab = np.bincount((lanes[a] >> 4) * 16 + (lanes[b] >> 4), minlength=256).astype(np.float64)
            chi = float(((tab - n / 256) ** 2 / (n / 256)).sum())
            assert chi < 255 + 5 * np.sqrt(2 * 255), (a, b, chi)
    # scale = 1 / P(keep): the mask has unit expectation for the rate actually drawn
    sm = rng.scaled_mask_attn(11, 5, rows, cols, rate, dtype=np.float64)
    assert abs(sm.mean() - 1.0) < 4 * np.sqrt((1 - p) / p / sm.size)
    assert (rng.scaled_mask_attn(1, 1, 4, 8, 0.0) == 1).all()
