"""Parity at BASELINE.json's full configuration (configs[1]: d256, 2+2, kernel_sizes [11,5,3], T384, F224, bf16) through
size-independent properties — the oracle cannot run this size in seconds, so the HIP path is checked against ITSELF
through identities that any correct implementation satisfies, and against the f32 HIP path (itself oracle-checked at
small sizes in test_model_gpu.py)."""
import numpy as np
import pytest
import torch

from ishara_amd import get_model

pytestmark = pytest.mark.gpu

KW = dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
          num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(384, 224))
B = 64


def _data(seed=1, b=B):
    g = np.random.default_rng(seed)
    x = g.standard_normal((b, 384, 224)).astype(np.float32)
    y = np.full((b, 64), 59, np.int64)
    for i in range(b):
        n = int(g.integers(8, 32))
        y[i, :n] = g.integers(0, 59, n)
    return torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()


@pytest.fixture(scope="module")
def models():
    mb = get_model(**KW, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
    mf = get_model(**KW, dropout_rate=0.2, dtype="f32", max_batch=B, seed=0)
    return mb, mf


def test_inference_is_per_sample_and_matches_f32(models):
    """Eval-mode forward: BatchNorm uses moving statistics, so every sample is independent — a batch of 64 must give the
    same logits as its two halves and as a permuted batch (up to GEMM tiling round-off), and bf16 must track f32."""
    mb, mf = models
    x, _ = _data(3)
    full = mb(x, training=False).float()
    halves = torch.cat([mb(x[:32], training=False), mb(x[32:], training=False)]).float()
    assert torch.isfinite(full).all()
    assert (full - halves).abs().max().item() <= 2e-2
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    assert (mb(x[perm], training=False).float() - full[perm]).abs().max().item() <= 2e-2
    f32 = mf(x, training=False)
    err = (full - f32).abs().max().item()
    assert err <= 0.15, f"bf16 vs f32 logits max-abs-err {err}"              # same tolerance as the oracle comparison
    # greedy decode at full size: the bf16 run's per-frame argmax equals the f32 run's on every frame the f32 logits resolve (top-2 margin
    # above 2x the observed difference) — counted, so that an empty comparison cannot pass
    from decode_check import frame_margins
    f32n, fulln = f32.cpu().numpy(), full.cpu().numpy()
    clear = frame_margins(f32n) > 2 * err
    assert clear.sum() >= 0.25 * clear.size, f"only {int(clear.sum())}/{clear.size} frames resolved at logit error {err:.3f}"
    assert (np.argmax(fulln, -1)[clear] == np.argmax(f32n, -1)[clear]).all()
    db, df = mb.decode_batch(full), mf.decode_batch(f32)
    whole = [i for i in range(B) if clear[i].all()]
    assert all(np.array_equal(db[i], df[i]) for i in whole)
    # decode is idempotent on its own logits
    assert all(np.array_equal(a, b) for a, b in zip(db, mb.decode_batch(full)))


def test_training_step_is_deterministic_and_learns(models):
    """Same seed -> bit-identical logits, loss AND gradients (no global float atomics in either pass: every cross-workgroup
    sum is a fixed-order reduction of partial rows); four steps on one batch lower the CTC loss; the gradients are linear in
    loss_scale."""
    mb, _ = models
    x, y = _data(5)
    w0 = mb.get_weights()
    l1, lg1 = mb.loss_and_gradients(x, y, seed=11); g1 = mb.grads.clone(); l1 = float(l1.item()); lg1 = lg1.clone()
    l2, lg2 = mb.loss_and_gradients(x, y, seed=11); g2 = mb.grads.clone(); l2 = float(l2.item())
    assert np.isfinite(l1) and l1 == l2 and torch.equal(lg1, lg2)
    ndiff = int((g1 != g2).sum().item())
    assert ndiff == 0, f"{ndiff} of {g1.numel()} gradient elements differ between identical runs (rel-L2 {((g1 - g2).norm() / g1.norm()).item():.3e})"
    mb.loss_and_gradients(x, y, seed=11, loss_scale=0.5)
    rel = ((mb.grads - 0.5 * g1).norm() / (0.5 * g1).norm()).item()
    assert rel <= 2e-3, f"gradients are not linear in loss_scale: {rel}"
    # the per-sample CTC loss of the training logits is a mean over samples: shuffling samples keeps the batch loss
    mb.optimizer.learning_rate = 1e-3
    losses = [float(mb.train_on_batch(x, y, seed=20 + i).item()) for i in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    mb.set_weights(w0)


def test_ctc_loss_is_permutation_equivariant(models):
    mb, _ = models
    x, y = _data(7)
    logits = mb(x, training=False)
    nll = mb.ctc_loss(y, logits)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    nll_p = mb.ctc_loss(y[perm], logits[perm])
    assert torch.allclose(nll_p, nll[perm], rtol=1e-6, atol=1e-4)
    assert (nll > 0).all() and torch.isfinite(nll).all()


def test_project_conv_wgrad_per_sample_affine_matches_materialised_operand(models):
    """Conv1DBlock backward, two routes: (a) the project conv's weight-gradient GEMM reads h2 and applies the per-sample affine
    (BatchNorm . ECA gate . drop-path) to its per-sample accumulators, emitting the BatchNorm / ECA statistics of dh4 on the way
    (gemm.hip TnPsa: 32 M-splits of 2 samples at B = 64); (b) ISHARA_NO_PSA: h4 is written by the forward pass, the GEMM reads it,
    and a separate pass over dh4 and h2 produces the statistics.  Same weights, batch and dropout seed: the logits are bit-identical
    (the forward pass differs only in what it stores) and the gradients agree to bf16 rounding of dh4 / h4."""
    import os
    mb, _ = models
    x, y = _data(9)
    os.environ["ISHARA_NO_PSA"] = "1"
    try:
        ma = get_model(**KW, dropout_rate=0.2, dtype="bf16", max_batch=B, seed=0)
    finally:
        del os.environ["ISHARA_NO_PSA"]
    ma.set_weights(mb.get_weights())
    lb, lgb = mb.loss_and_gradients(x, y, seed=13); gb = mb.grads.clone(); lgb = lgb.clone()
    la, lga = ma.loss_and_gradients(x, y, seed=13); ga = ma.grads.clone()
    assert torch.equal(lga, lgb) and float(la.item()) == float(lb.item())
    rel = ((ga - gb).norm() / ga.norm()).item()
    assert rel <= 2e-2, f"flat gradient rel-L2 between the two routes {rel}"
    worst = 0.0
    worst_name = ""
    for name, shape, off, trainable in mb.entries:
        if not trainable or name.endswith("/depthwise_conv/bias"):
            continue        # (the Conformer conv's depthwise bias feeds a BatchNorm: its exact gradient is 0, what is computed is rounding residue)
        n = int(np.prod(shape))
        a, b_ = ga[off:off + n], gb[off:off + n]
        den = a.norm().item()
        if den > 0:
            e = (a - b_).norm().item() / den
            if e > worst:
                worst, worst_name = e, name
    assert worst <= 6e-2, f"worst per-parameter rel-L2 {worst} ({worst_name})"


def test_project_conv_wgrad_psa_config4_shape():
    """The same two-route comparison for configs[3]'s shapes (d512: the project conv is K = 1024 -> N = 512, 32 output tiles, T = 512 = 16
    stages per sample, two samples per M-split at B = 16; the finalize kernel works on four 256-channel chunks per sample)."""
    import os
    kw = dict(dim=512, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
              num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(512, 224))
    b = 16
    g = np.random.default_rng(21)
    x = torch.from_numpy(g.standard_normal((b, 512, 224)).astype(np.float32)).cuda()
    y = np.full((b, 64), 59, np.int64)
    for i in range(b):
        n = int(g.integers(8, 32))
        y[i, :n] = g.integers(0, 59, n)
    y = torch.from_numpy(y).cuda()
    mb = get_model(**kw, dropout_rate=0.2, dtype="bf16", max_batch=b, seed=0)
    os.environ["ISHARA_NO_PSA"] = "1"
    try:
        ma = get_model(**kw, dropout_rate=0.2, dtype="bf16", max_batch=b, seed=0)
    finally:
        del os.environ["ISHARA_NO_PSA"]
    ma.set_weights(mb.get_weights())
    lb, lgb = mb.loss_and_gradients(x, y, seed=5); gb = mb.grads.clone(); lgb = lgb.clone()
    la, lga = ma.loss_and_gradients(x, y, seed=5); ga = ma.grads.clone()
    assert torch.equal(lga, lgb) and float(la.item()) == float(lb.item())
    assert torch.isfinite(gb).all()
    rel = ((ga - gb).norm() / ga.norm()).item()
    assert rel <= 2e-2, f"flat gradient rel-L2 between the two routes {rel}"
    worst, worst_name = 0.0, ""
    for name, shape, off, trainable in mb.entries:
        if not trainable or name.endswith("/depthwise_conv/bias"):
            continue
        n = int(np.prod(shape))
        a, b_ = ga[off:off + n], gb[off:off + n]
        den = a.norm().item()
        if den > 0:
            e = (a - b_).norm().item() / den
            if e > worst:
                worst, worst_name = e, name
    assert worst <= 6e-2, f"worst per-parameter rel-L2 {worst} ({worst_name})"


def test_config4_big_tile_gemms_match_small_tile_route():
    """configs[3]'s shapes at M = B * T = 32768 rows, where the 256 x 256 tile GEMMs of gemm_big.hip take over the dense layers with K, N >= 512
    (forward / dgrad: gemm_nt_big_kernel; weight gradients incl. the drop-path weighted bias sums: gemm_tn_big_kernel): the same model, batch
    and dropout seed with the big kernels switched off (ishara_debug_set_nt_big(0): A-stationary + 128 x 128 tile kernels).  The forward
    accumulations run in the same order over K, so loss and logits agree to bf16 rounding of single outputs; the weight gradients split M
    differently (fp32 summation order)."""
    from ishara_amd import _lib
    lib = _lib.load()
    kw = dict(dim=512, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
              num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(512, 224))
    b = 64
    g = np.random.default_rng(33)
    x = torch.from_numpy(g.standard_normal((b, 512, 224)).astype(np.float32)).cuda()
    y = np.full((b, 64), 59, np.int64)
    for i in range(b):
        n = int(g.integers(8, 32))
        y[i, :n] = g.integers(0, 59, n)
    y = torch.from_numpy(y).cuda()
    m = get_model(**kw, dropout_rate=0.2, dtype="bf16", max_batch=b, seed=0)
    res = {}
    try:
        for on in (0, 1):
            lib.ishara_debug_set_nt_big(on)
            l, lg = m.loss_and_gradients(x, y, seed=5)
            res[on] = (float(l.item()), lg.clone().float(), m.grads.clone())
    finally:
        lib.ishara_debug_set_nt_big(1)
    (l0, lg0, g0), (l1, lg1, g1) = res[0], res[1]
    assert abs(l0 - l1) <= 2e-3 * abs(l0), (l0, l1)
    assert (lg0 - lg1).abs().max().item() <= 0.15, (lg0 - lg1).abs().max().item()          # logits: a few bf16 roundings through 2 blocks
    assert torch.isfinite(g1).all()
    rel = ((g0 - g1).norm() / g0.norm()).item()
    assert rel <= 2e-2, f"flat gradient rel-L2 between the two routes {rel}"
    worst, worst_name = 0.0, ""
    for name, shape, off, trainable in m.entries:
        if not trainable or name.endswith("/depthwise_conv/bias"):
            continue
        n = int(np.prod(shape))
        a, b_ = g0[off:off + n], g1[off:off + n]
        den = a.norm().item()
        if den > 0:
            e = (a - b_).norm().item() / den
            if e > worst:
                worst, worst_name = e, name
    assert worst <= 6e-2, f"worst per-parameter rel-L2 {worst} ({worst_name})"
