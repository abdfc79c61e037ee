"""CPU oracle for the Ishara hot path.

TEST INFRASTRUCTURE ONLY.  This module is a from-scratch CPU restatement
(torch-CPU, fp32 or fp64) of the TensorFlow/Keras model the reference defines
in `Test Notebooks/conv-hybrid-model.ipynb` (cells c5-c8, c11).  It exists to
check the HIP path in ishara_amd/; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product path never does.

PARITY STATUS: **parity unpinned** for numerical values.  The reference's
arithmetic lives in tensorflow / tensorflow-addons (unpinned versions,
Dockerfile:15-21; absent from this image) and the reference holds no golden
vector or test for this path.  The structural pins that do
exist (saved `model.summary()` parameter counts, tensor shapes) are checked in
tests/test_oracle_pins.py against tests/golden/structural_pins.json.  CTC is
cross-checked against torch.nn.functional.ctc_loss (an independent
implementation of the same published algorithm).

Citations `cN:L` are to conv-hybrid-model.ipynb, code cell N, line L.
Keras conventions restated here: Dense kernel [in,out]; DepthwiseConv1D kernel
[k,C]; BatchNormalization eps=1e-3, biased batch variance, moving = moving*m +
batch*(1-m); LayerNormalization eps as given (default 1e-3); inverted dropout.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import rng

BLANK = 59  # pad_token_idx, c1:5
BN_EPS = 1e-3  # Keras BatchNormalization default epsilon


@dataclass
class Config:
    """get_model kwargs (c7:1-11) + the notebook globals it closes over."""
    dim: int = 256
    num_conv_squeeze_blocks: int = 2
    num_conv_conform_blocks: int = 2
    kernel_sizes: Sequence[int] = (11, 5, 3)
    num_conv_per_block: int = 3
    dropout_rate: float = 0.2
    num_heads: int = 8
    expansion_factor: int = 2
    transformer_kernel_size: int = 15
    input_shape: Tuple[int, int] = (176, 276)   # INPUT_SHAPE, c3:119
    num_classes: int = 60                        # len(char_to_num), c1:7
    # knobs needed only to restate the *variant* notebooks whose saved
    # summaries pin the parameter counts; defaults = c7.
    top_dim: Optional[int] = None                # c7:61 -> 2*dim
    squeeze_expansion: Optional[int] = None
    conformer_expansion: Optional[int] = None
    head_dropout: float = 0.4                    # c7:62
    conformer_attn_dropout: float = 0.1          # c5:312 default (not passed at c7:51-58)

    @property
    def T(self): return self.input_shape[0]
    @property
    def F(self): return self.input_shape[1]
    @property
    def e_sq(self): return self.squeeze_expansion or self.expansion_factor
    @property
    def e_cf(self): return self.conformer_expansion or self.expansion_factor
    @property
    def dtop(self): return self.top_dim or 2 * self.dim


# ----------------------------------------------------------------------------
# Parameter inventory (row 1).  Order = Keras layer creation order.
# entry: (name, shape, init, trainable)
# ----------------------------------------------------------------------------
def _conv1d_block_specs(name, d, k):
    c = 2 * d  # expand_ratio=2, c5:45,57
    return [
        (f"{name}_expand_conv/kernel", (d, c), "glorot", True),
        (f"{name}_expand_conv/bias", (c,), "zeros", True),
        (f"{name}_dwconv/depthwise_kernel", (k, c), "glorot_dw", True),
        (f"{name}_bn/gamma", (c,), "ones", True),
        (f"{name}_bn/beta", (c,), "zeros", True),
        (f"{name}_bn/moving_mean", (c,), "zeros", False),
        (f"{name}_bn/moving_variance", (c,), "ones", False),
        (f"{name}_eca/kernel", (5,), "glorot_eca", True),
        (f"{name}_project_conv/kernel", (c, d), "glorot", True),
        (f"{name}_project_conv/bias", (d,), "zeros", True),
    ]


def _ln(name, d):
    return [(f"{name}/gamma", (d,), "ones", True), (f"{name}/beta", (d,), "zeros", True)]


def _dense(name, i, o, bias=True):
    s = [(f"{name}/kernel", (i, o), "glorot", True)]
    if bias:
        s.append((f"{name}/bias", (o,), "zeros", True))
    return s


def _squeezeformer_specs(name, d, e, k):
    r = max(1, d // 8)  # c5:125
    s = []
    s += _ln(f"{name}/norm1", d)
    s += _dense(f"{name}/ffn1_dense1", d, d * e) + _dense(f"{name}/ffn1_dense2", d * e, d)
    s += _ln(f"{name}/norm2", d)
    s += _dense(f"{name}/mha/qkv", d, 3 * d, bias=False) + _dense(f"{name}/mha/proj", d, d, bias=False)
    s += _ln(f"{name}/conv/norm", d)
    s += _dense(f"{name}/conv/conv1", d, d * e)
    s += [(f"{name}/conv/conv2/depthwise_kernel", (k, d * e), "glorot_dw", True)]
    s += _dense(f"{name}/conv/conv3", d * e, d)
    s += _dense(f"{name}/conv/se/fc1", d, r) + _dense(f"{name}/conv/se/fc2", r, d)
    s += _ln(f"{name}/norm3", d)
    s += _dense(f"{name}/ffn2_dense1", d, d * e) + _dense(f"{name}/ffn2_dense2", d * e, d)
    return s


def _conformer_specs(name, d, e, k):
    s = []
    s += _dense(f"{name}/ffn1/dense1", d, d * e) + _dense(f"{name}/ffn1/dense2", d * e, d)
    s += _dense(f"{name}/mha/qkv", d, 3 * d, bias=False) + _dense(f"{name}/mha/proj", d, d, bias=False)
    s += _dense(f"{name}/conv/pointwise_conv1", d, 2 * d)
    s += [(f"{name}/conv/depthwise_conv/kernel", (k, d), "glorot_dw", True),
          (f"{name}/conv/depthwise_conv/bias", (d,), "zeros", True)]
    s += _dense(f"{name}/conv/pointwise_conv2", d, d)
    s += [(f"{name}/conv/batch_norm/gamma", (d,), "ones", True),
          (f"{name}/conv/batch_norm/beta", (d,), "zeros", True),
          (f"{name}/conv/batch_norm/moving_mean", (d,), "zeros", False),
          (f"{name}/conv/batch_norm/moving_variance", (d,), "ones", False)]
    s += _ln(f"{name}/conv/layer_norm", d)
    s += _dense(f"{name}/ffn2/dense1", d, d * e) + _dense(f"{name}/ffn2/dense2", d * e, d)
    s += _ln(f"{name}/layer_norm1", d) + _ln(f"{name}/layer_norm2", d)
    return s


def param_specs(cfg: Config) -> List[tuple]:
    d = cfg.dim
    s = [("stem_conv/kernel", (cfg.F, d), "glorot", True),
         ("stem_bn/gamma", (d,), "ones", True), ("stem_bn/beta", (d,), "zeros", True),
         ("stem_bn/moving_mean", (d,), "zeros", False), ("stem_bn/moving_variance", (d,), "ones", False)]

    def conv_blocks(tag):
        out = []
        for j in range(cfg.num_conv_per_block):
            k = cfg.kernel_sizes[j % len(cfg.kernel_sizes)]       # c7:22
            out += _conv1d_block_specs(f"conv{tag}_{j + 1}", d, k)  # c7:27
        return out

    for i in range(cfg.num_conv_squeeze_blocks):
        s += conv_blocks(f"squeeze_{i}")
        s += _squeezeformer_specs(f"squeezeformer_{i}", d, cfg.e_sq, cfg.transformer_kernel_size)
    for i in range(cfg.num_conv_conform_blocks):
        s += conv_blocks(f"conform_{i}")
        s += _conformer_specs(f"conformer_{i}", d, cfg.e_cf, cfg.transformer_kernel_size)
    s += _dense("top_conv", d, cfg.dtop) + _dense("classifier", cfg.dtop, cfg.num_classes)
    return s


def count_params(cfg: Config):
    tr = sum(int(np.prod(sh)) for _, sh, _, t in param_specs(cfg) if t)
    nt = sum(int(np.prod(sh)) for _, sh, _, t in param_specs(cfg) if not t)
    return tr + nt, tr, nt


def init_params(cfg: Config, seed: int = 0) -> Dict[str, np.ndarray]:
    """Keras default initialisers (glorot_uniform kernels, zero biases, (1,0)
    norms, BN moving (0,1)); fans per keras.initializers._compute_fans."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape, init, _ in param_specs(cfg):
        if init == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            w = g.uniform(-lim, lim, size=shape)
        elif init == "glorot_dw":      # depthwise kernel (k, C, 1): fan_in=k*C, fan_out=k
            k, c = shape
            lim = math.sqrt(6.0 / (k * c + k))
            w = g.uniform(-lim, lim, size=shape)
        elif init == "glorot_eca":     # Conv1D(1, k=5) kernel (5,1,1): fans 5,5
            lim = math.sqrt(6.0 / 10.0)
            w = g.uniform(-lim, lim, size=shape)
        elif init == "ones":
            w = np.ones(shape)
        else:
            w = np.zeros(shape)
        out[name] = w.astype(np.float32)
    return out


# ----------------------------------------------------------------------------
# Dropout sites (deterministic order shared with ishara_amd/csrc/model.cpp)
# ----------------------------------------------------------------------------
class _Sites:
    def __init__(self, seed, training):
        self.seed, self.training, self.n = seed, training, 0

    def mask(self, rows, cols, rate, ref: torch.Tensor, attn: bool = False):
        """Returns multiplicative mask tensor [rows, cols] or None; always
        consumes one site id so ids stay aligned with the HIP library.  attn: an attention-probability
        site (one hash per four keys, 8-bit threshold: oracle/rng.py keep_mask_attn)."""
        site = self.n
        self.n += 1
        if not self.training or rate <= 0.0:
            return None
        m = (rng.scaled_mask_attn if attn else rng.scaled_mask)(self.seed, site, rows, cols, rate, dtype=np.float64)
        return torch.from_numpy(m).to(ref.dtype)


# ----------------------------------------------------------------------------
# Layers
# ----------------------------------------------------------------------------
def swish(x):
    return x * torch.sigmoid(x)


def positional_encoding(maxlen, num_hid, dtype=torch.float32):
    """c5:226-235 — concatenated [sin | cos] halves, computed in fp32 like TF."""
    depth = num_hid / 2
    positions = torch.arange(maxlen, dtype=torch.float32)[:, None]
    depths = torch.arange(int(depth), dtype=torch.float32)[None, :] / depth
    angle_rates = 1.0 / torch.pow(torch.tensor(10000.0, dtype=torch.float32), depths)
    angle_rads = positions @ angle_rates
    return torch.cat([torch.sin(angle_rads), torch.cos(angle_rads)], dim=-1).to(dtype)


def batch_norm(x, P, name, training, momentum, new_stats):
    """Keras BatchNormalization over the last axis of [B,T,C]."""
    g, b = P[f"{name}/gamma"], P[f"{name}/beta"]
    if training:
        mean = x.mean(dim=(0, 1))
        var = ((x - mean) ** 2).mean(dim=(0, 1))          # biased
        new_stats[f"{name}/moving_mean"] = (P[f"{name}/moving_mean"] * momentum + mean * (1 - momentum)).detach()
        new_stats[f"{name}/moving_variance"] = (P[f"{name}/moving_variance"] * momentum + var * (1 - momentum)).detach()
    else:
        mean, var = P[f"{name}/moving_mean"], P[f"{name}/moving_variance"]
    return (x - mean) * torch.rsqrt(var + BN_EPS) * g + b


def layer_norm(x, P, name, eps):
    return F.layer_norm(x, (x.shape[-1],), P[f"{name}/gamma"], P[f"{name}/beta"], eps)


def dense(x, P, name, bias=True):
    y = x @ P[f"{name}/kernel"]
    return y + P[f"{name}/bias"] if bias else y


def causal_dwconv(x, w):
    """c5:17-39: left zero-pad k-1, depthwise 'valid'.  x [B,T,C], w [k,C]."""
    k = w.shape[0]
    xp = F.pad(x.transpose(1, 2), (k - 1, 0))
    return F.conv1d(xp, w.t().unsqueeze(1), groups=w.shape[1]).transpose(1, 2)


def same_dwconv(x, w, b):
    """c5:265-271: Conv1D(groups=C, padding='same'), odd k, with bias."""
    k = w.shape[0]
    xp = F.pad(x.transpose(1, 2), ((k - 1) // 2, k // 2))
    return F.conv1d(xp, w.t().unsqueeze(1), b, groups=w.shape[1]).transpose(1, 2)


def eca(x, w):
    """c5:1-15: GAP over T -> Conv1D(1,k=5,'same',no bias) over the channel axis."""
    g = x.mean(dim=1)                                      # [B,C]
    k = w.shape[0]
    z = F.conv1d(g.unsqueeze(1), w.view(1, 1, k), padding=(k - 1) // 2).squeeze(1)
    return x * torch.sigmoid(z)[:, None, :]


def conv1d_block(x, P, name, cfg, training, sites, new_stats):
    """c5:41-89."""
    skip = x
    h = swish(dense(x, P, f"{name}_expand_conv"))
    h = causal_dwconv(h, P[f"{name}_dwconv/depthwise_kernel"])
    h = batch_norm(h, P, f"{name}_bn", training, 0.95, new_stats)
    h = eca(h, P[f"{name}_eca/kernel"])
    h = dense(h, P, f"{name}_project_conv")
    m = sites.mask(x.shape[0], 1, cfg.dropout_rate, x)     # noise_shape=(None,1,1), c5:83
    if m is not None:
        h = h * m[:, :, None]
    return h + skip                                        # c5:85-86


def mhsa(x, P, name, cfg, rate, sites):
    """c5:91-118; head-major qkv packing, scale = dim**-0.5 (c5:95)."""
    B, T, d = x.shape
    H = cfg.num_heads
    dh = d // H
    qkv = dense(x, P, f"{name}/qkv", bias=False).view(B, T, H, 3 * dh).permute(0, 2, 1, 3)
    q, k, v = qkv[..., :dh], qkv[..., dh:2 * dh], qkv[..., 2 * dh:]
    attn = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    m = sites.mask(B * H * T, T, rate, x, attn=True)
    if m is not None:
        attn = attn * m.view(B, H, T, T)
    o = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    return dense(o, P, f"{name}/proj", bias=False)


def _drop(x, rate, sites):
    B, T, C = x.shape
    m = sites.mask(B * T, C, rate, x)
    return x if m is None else x * m.view(B, T, C)


def ffn(x, P, n1, n2, rate, sites):
    h = swish(dense(x, P, n1))
    h = _drop(h, rate, sites)
    return dense(h, P, n2)


def squeezeformer_block(x, P, name, cfg, training, sites):
    """c5:155-207 (+ConvModule c5:135-153, SqueezeExcite c5:120-133)."""
    r = cfg.dropout_rate
    x = x + _drop(ffn(layer_norm(x, P, f"{name}/norm1", 1e-6), P, f"{name}/ffn1_dense1", f"{name}/ffn1_dense2", r, sites), r, sites)
    x = x + _drop(mhsa(layer_norm(x, P, f"{name}/norm2", 1e-6), P, f"{name}/mha", cfg, r, sites), r, sites)
    # ConvModule
    u = layer_norm(x, P, f"{name}/conv/norm", 1e-6)
    u = swish(dense(u, P, f"{name}/conv/conv1"))
    u = swish(causal_dwconv(u, P[f"{name}/conv/conv2/depthwise_kernel"]))
    u = dense(u, P, f"{name}/conv/conv3")
    z = u.mean(dim=1)
    z = swish(dense(z, P, f"{name}/conv/se/fc1"))
    z = torch.sigmoid(dense(z, P, f"{name}/conv/se/fc2"))
    x = u * z[:, None, :] + x
    x = x + _drop(ffn(layer_norm(x, P, f"{name}/norm3", 1e-6), P, f"{name}/ffn2_dense1", f"{name}/ffn2_dense2", r, sites), r, sites)
    return x


def conformer_block(x, P, name, cfg, training, sites, new_stats):
    """c5:311-343 (+FeedForwardModule c5:237-247, ConvolutionModule c5:249-309)."""
    r = cfg.dropout_rate
    x = x + ffn(layer_norm(x, P, f"{name}/layer_norm1", 1e-6), P, f"{name}/ffn1/dense1", f"{name}/ffn1/dense2", r, sites)
    x = x + mhsa(layer_norm(x, P, f"{name}/layer_norm1", 1e-6), P, f"{name}/mha", cfg, cfg.conformer_attn_dropout, sites)
    res = x
    u = dense(x, P, f"{name}/conv/pointwise_conv1")
    d = x.shape[-1]
    u = u[..., :d] * torch.sigmoid(u[..., d:])             # c5:294-295
    u = same_dwconv(u, P[f"{name}/conv/depthwise_conv/kernel"], P[f"{name}/conv/depthwise_conv/bias"])
    u = batch_norm(u, P, f"{name}/conv/batch_norm", training, 0.99, new_stats)
    u = dense(u, P, f"{name}/conv/pointwise_conv2")
    x = layer_norm(u + res, P, f"{name}/conv/layer_norm", 1e-3)
    x = x + ffn(layer_norm(x, P, f"{name}/layer_norm2", 1e-6), P, f"{name}/ffn2/dense1", f"{name}/ffn2/dense2", r, sites)
    return x


def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, cfg: Config, training: bool = False,
            seed: int = 0, taps: Optional[dict] = None):
    """get_model(...)(x) — c7:12-65.  Returns (logits [B,T,C], new_bn_stats).

    Masking(0.0) (c7:13) is numerically a no-op here: an all-zero frame maps
    to a zero row through the bias-free stem Dense, and the mask is dropped at
    the `x + pe` TFOpLambda."""
    sites = _Sites(seed, training)
    new_stats: Dict[str, torch.Tensor] = {}
    B, T, Fdim = x.shape
    h = x @ P["stem_conv/kernel"]
    h = h + positional_encoding(T, cfg.dim, h.dtype)
    h = batch_norm(h, P, "stem_bn", training, 0.95, new_stats)
    if taps is not None: taps["stem"] = h

    def conv_blocks(h, tag):
        for j in range(cfg.num_conv_per_block):
            h = conv1d_block(h, P, f"conv{tag}_{j + 1}", cfg, training, sites, new_stats)
            if taps is not None: taps[f"conv{tag}_{j + 1}"] = h
        return h

    for i in range(cfg.num_conv_squeeze_blocks):
        h = conv_blocks(h, f"squeeze_{i}")
        h = squeezeformer_block(h, P, f"squeezeformer_{i}", cfg, training, sites)
        if taps is not None: taps[f"squeezeformer_{i}"] = h
    for i in range(cfg.num_conv_conform_blocks):
        h = conv_blocks(h, f"conform_{i}")
        h = conformer_block(h, P, f"conformer_{i}", cfg, training, sites, new_stats)
        if taps is not None: taps[f"conformer_{i}"] = h
    h = torch.relu(dense(h, P, "top_conv"))
    h = _drop(h, cfg.head_dropout, sites)
    return dense(h, P, "classifier"), new_stats


# ----------------------------------------------------------------------------
# CTC loss (row 8) — differentiable log-space alpha recursion
# ----------------------------------------------------------------------------
def ctc_nll(labels: torch.Tensor, logits: torch.Tensor, blank: int = BLANK) -> torch.Tensor:
    """Per-sample -log p(y|x) of tf.nn.ctc_loss(labels, logits, label_length,
    logit_length=T, blank_index=59, logits_time_major=False) as called at
    c6:1-11: label_length = #(labels != 59); log-softmax applied internally."""
    B, T, C = logits.shape
    lp = torch.log_softmax(logits, dim=-1)
    lab_len = (labels != blank).sum(dim=1)
    L = labels.shape[1]
    S = 2 * L + 1
    ext = torch.full((B, S), blank, dtype=torch.long)
    ext[:, 1::2] = labels
    NEG = torch.tensor(-1e30, dtype=logits.dtype)
    # transitions s-2 -> s allowed when ext[s] != blank and ext[s] != ext[s-2]
    allow2 = torch.zeros((B, S), dtype=torch.bool)
    allow2[:, 2:] = (ext[:, 2:] != blank) & (ext[:, 2:] != ext[:, :-2])
    valid = torch.arange(S)[None, :] < (2 * lab_len[:, None] + 1)
    alpha = torch.full((B, S), -1e30, dtype=logits.dtype)
    e0 = lp[:, 0, :].gather(1, ext)
    init = torch.zeros((B, S), dtype=torch.bool)
    init[:, 0] = True
    init[:, 1] = lab_len > 0
    alpha = torch.where(init & valid, e0, alpha)
    for t in range(1, T):
        a1 = torch.cat([NEG.expand(B, 1), alpha[:, :-1]], dim=1)
        a2 = torch.cat([NEG.expand(B, 2), alpha[:, :-2]], dim=1)
        a2 = torch.where(allow2, a2, NEG)
        m = torch.maximum(torch.maximum(alpha, a1), a2)
        s = torch.exp(alpha - m) + torch.exp(a1 - m) + torch.exp(a2 - m)
        new = m + torch.log(s) + lp[:, t, :].gather(1, ext)
        alpha = torch.where(valid, new, NEG)
    last = 2 * lab_len                                      # final blank
    aL = alpha.gather(1, last[:, None]).squeeze(1)
    aLm1 = alpha.gather(1, (last - 1).clamp(min=0)[:, None]).squeeze(1)
    aLm1 = torch.where(lab_len > 0, aLm1, NEG)
    m = torch.maximum(aL, aLm1)
    return -(m + torch.log(torch.exp(aL - m) + torch.exp(aLm1 - m)))


def ctc_loss(labels, logits):
    """CTCLoss, c6:1-13: mean over the batch."""
    return ctc_nll(labels, logits).mean()


# ----------------------------------------------------------------------------
# Greedy decode (row 9)
# ----------------------------------------------------------------------------
def decode_phrase(pred: np.ndarray, blank: int = BLANK) -> np.ndarray:
    """c8:4-12.  pred [T,C].  Keeps x[i] (i <= T-2) where x[i] != x[i+1], then
    drops blanks — the final run is never emitted (reference quirk)."""
    x = np.argmax(pred, axis=1)                # first max index on ties
    keep = np.nonzero(x[:-1] != x[1:])[0]
    x = x[keep]
    return x[x != blank].astype(np.int64)


FALLBACK_PHRASE = np.array([17, 0, 32, 12, 36, 0, 12, 32, 49, 46, 36], dtype=np.int64)  # c13:22-23


def tflite_postprocess(idx: np.ndarray) -> np.ndarray:
    """c13:22-24: len<3 fallback then one_hot(., 59) (index 59 -> zero row)."""
    if idx.shape[0] < 3:
        idx = FALLBACK_PHRASE
    out = np.zeros((idx.shape[0], 59), dtype=np.float32)
    ok = idx < 59
    out[np.arange(idx.shape[0])[ok], idx[ok]] = 1.0
    return out


# ----------------------------------------------------------------------------
# Optimizer (row 10): tfa RectifiedAdam(sma_threshold=4) inside Lookahead(5, 0.5)
# Source is tensorflow-addons (not in the reference repo) -> parity unpinned;
# restated from the published algorithm (Liu et al. 2019; Zhang et al. 2019)
# with tfa's defaults: beta1 .9, beta2 .999, eps 1e-7, weight_decay 0.
# ----------------------------------------------------------------------------
@dataclass
class OptState:
    m: np.ndarray
    v: np.ndarray
    slow: np.ndarray
    step: int = 0


def radam_coeffs(step, beta1=0.9, beta2=0.999, sma_threshold=4.0):
    """Host-side scalars for step `step` (1-based)."""
    b1p, b2p = beta1 ** step, beta2 ** step
    sma_inf = 2.0 / (1.0 - beta2) - 1.0
    sma_t = sma_inf - 2.0 * step * b2p / (1.0 - b2p)
    rect = sma_t >= sma_threshold
    r_t = math.sqrt(max((sma_t - 4.0) / (sma_inf - 4.0) * (sma_t - 2.0) / (sma_inf - 2.0) * sma_inf / sma_t, 0.0)) if rect else 0.0
    return dict(c1=1.0 / (1.0 - b1p), c2=1.0 / (1.0 - b2p), r_t=r_t, rect=bool(rect))


def optimizer_init(theta: np.ndarray) -> OptState:
    return OptState(np.zeros_like(theta), np.zeros_like(theta), theta.copy(), 0)


def optimizer_step(theta, grad, st: OptState, lr, weight_decay=0.0, beta1=0.9, beta2=0.999,
                   eps=1e-7, sync_period=5, slow_step=0.5):
    st.step += 1
    c = radam_coeffs(st.step, beta1, beta2)
    st.m = beta1 * st.m + (1 - beta1) * grad
    st.v = beta2 * st.v + (1 - beta2) * grad * grad
    m_hat = st.m * np.float32(c["c1"])
    if c["rect"]:
        upd = np.float32(c["r_t"]) * m_hat / (np.sqrt(st.v * np.float32(c["c2"])) + np.float32(eps))
    else:
        upd = m_hat
    if weight_decay:
        upd = upd + np.float32(weight_decay) * theta
    theta = theta - np.float32(lr) * upd
    if st.step % sync_period == 0:
        st.slow = st.slow + np.float32(slow_step) * (theta - st.slow)
        theta = st.slow.copy()
    return theta


def lrfn(current_step, num_warmup_steps, lr_max, num_cycles=0.50, num_training_steps=50, warmup_method="exp"):
    """c11:1-11."""
    if current_step < num_warmup_steps:
        if warmup_method == "log":
            return lr_max * 0.10 ** (num_warmup_steps - current_step)
        return lr_max * 2 ** -(num_warmup_steps - current_step)
    progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))) * lr_max


# ----------------------------------------------------------------------------
# Convenience: one full training step on CPU (used as cpu_baseline "port")
# ----------------------------------------------------------------------------
def to_torch(params: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        if requires_grad and "moving_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def loss_and_grads(params: Dict[str, np.ndarray], x: np.ndarray, y: np.ndarray, cfg: Config,
                   training=True, seed=0, dtype=torch.float32):
    P = to_torch(params, dtype)
    xt = torch.from_numpy(x).to(dtype)
    logits, new_stats = forward(P, xt, cfg, training=training, seed=seed)
    loss = ctc_loss(torch.from_numpy(y).long(), logits)
    loss.backward()
    grads = {k: (v.grad.detach().numpy() if v.grad is not None else None) for k, v in P.items() if v.requires_grad}
    return float(loss.detach()), logits.detach().numpy(), grads, {k: v.numpy() for k, v in new_stats.items()}


def synthetic_batch(cfg: Config, B: int, seed: int = 1):
    """SURVEY §8d synthetic inputs: x ~ N(0,1); labels length U[8,31] over
    {0..58} padded to 64 with 59 (clamped so the CTC path is feasible)."""
    g = np.random.default_rng(seed)
    x = g.standard_normal((B, cfg.T, cfg.F)).astype(np.float32)
    y = np.full((B, 64), BLANK, dtype=np.int64)
    hi = max(2, min(31, (cfg.T - 1) // 2))
    lo = min(8, hi)
    for b in range(B):
        n = int(g.integers(lo, hi + 1))
        y[b, :n] = g.integers(0, 59, size=n)
    return x, y
