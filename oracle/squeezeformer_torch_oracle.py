"""CPU restatement of the reference's torch Squeezeformer family — TEST INFRASTRUCTURE ONLY.

PARITY PINNED by vectors produced by running the reference's own files: oracle/gen_golden_squeezeformer.py loads
`/root/reference/squeezeformer/{attention,modules,convolution,encoder}.py` in the build container (the package `__init__` is bypassed
and the missing `squeezeformer.activation.Swish` is satisfied by the `Swish` class convolution.py:22-27 itself defines) and writes
tests/golden/squeezeformer_r4.npz; tests/test_golden_squeezeformer.py checks every function below against it (sub-module outputs,
per-layer outputs, training-mode output, autograd gradients, BatchNorm statistics).  It follows the files line by line:

  R1  RelativeMultiHeadAttention            squeezeformer/attention.py:25-110   (`_relative_shift` :102-110)
      MultiHeadedSelfAttentionModule         attention.py:113-139
  R2  RelPositionalEncoding                  squeezeformer/modules.py:59-108
  R3  ConvModule (+ Swish, GLU, DW / PW conv) squeezeformer/convolution.py:199-238 (:22-37, :117-196)
  R4  FeedForwardModule, ResidualConnectionModule, recover_resolution      modules.py:24-56, 111-123, 137-142
      DepthwiseConv2dSubsampling, TimeReductionLayer                      convolution.py:39-73, 241-269
      SqueezeformerBlock (post-LN), SqueezeformerEncoder                  encoder.py:169-247, 26-166

Parameters are a flat dict keyed by the names the torch modules would give (`layers.3.module.sequential.0.module.attention.
query_proj.weight`, ...; a `ResidualConnectionModule` adds `.module`), arrays in torch layout.  Pure functional torch, any dtype;
dropout off (p = 0) unless a mask callback is given.
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------ R2
def rel_positional_encoding(T: int, d: int, dtype=torch.float32) -> torch.Tensor:
    """modules.py:73-108 for an input of T frames: [1, 2T-1, d]; row r holds relative position T-1-r (positive first:
    `pe_positive` flipped, then `pe_negative[1:]`); even columns sin, odd columns cos."""
    position = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pos = torch.zeros(T, d)
    neg = torch.zeros(T, d)
    pos[:, 0::2] = torch.sin(position * div_term)
    pos[:, 1::2] = torch.cos(position * div_term)
    neg[:, 0::2] = torch.sin(-1 * position * div_term)
    neg[:, 1::2] = torch.cos(-1 * position * div_term)
    pe = torch.cat([torch.flip(pos, [0]).unsqueeze(0), neg[1:].unsqueeze(0)], dim=1)
    return pe.to(dtype)


# ------------------------------------------------------------------------------------------------ R1
def relative_shift(pos_score: torch.Tensor) -> torch.Tensor:
    """attention.py:102-110, the Transformer-XL pad / reshape / slice."""
    B, H, T1, T2 = pos_score.shape
    zeros = pos_score.new_zeros(B, H, T1, 1)
    padded = torch.cat([zeros, pos_score], dim=-1).view(B, H, T2 + 1, T1)
    return padded[:, :, 1:].reshape(B, H, T1, T2)[:, :, :, : T2 // 2 + 1]


def relative_shift_closed_form(pos_score: torch.Tensor) -> torch.Tensor:
    """The same map as an index formula (what the HIP kernel computes): out[i, j] = pos[i, T-1-i+j] for T2 = 2T-1."""
    B, H, T, T2 = pos_score.shape
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    return pos_score[:, :, i, T - 1 - i + j]


def rel_mhsa(x, P, p, heads, drop=None):
    """MultiHeadedSelfAttentionModule.forward (attention.py:130-139) around RelativeMultiHeadAttention.forward (:69-100).
    `drop(attn, site)` applies a dropout mask when given (sites: 0 attention probabilities, 1 module output)."""
    B, T, d = x.shape
    dh = d // heads
    a = p + ".attention"
    pe = rel_positional_encoding(T, d, x.dtype).repeat(B, 1, 1)
    q = (x @ P[a + ".query_proj.weight"].t() + P[a + ".query_proj.bias"]).view(B, T, heads, dh)
    k = (x @ P[a + ".key_proj.weight"].t() + P[a + ".key_proj.bias"]).view(B, T, heads, dh).permute(0, 2, 1, 3)
    v = (x @ P[a + ".value_proj.weight"].t() + P[a + ".value_proj.bias"]).view(B, T, heads, dh).permute(0, 2, 1, 3)
    pos = (pe @ P[a + ".pos_proj.weight"].t()).view(B, -1, heads, dh)
    content = torch.matmul((q + P[a + ".u_bias"]).transpose(1, 2), k.transpose(2, 3))
    pos_score = relative_shift(torch.matmul((q + P[a + ".v_bias"]).transpose(1, 2), pos.permute(0, 2, 3, 1)))
    attn = torch.softmax((content + pos_score) / math.sqrt(dh), -1)
    if drop is not None:
        attn = drop(attn, 0)
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, d)
    out = ctx @ P[a + ".out_proj.weight"].t() + P[a + ".out_proj.bias"]
    return drop(out, 1) if drop is not None else out


# ------------------------------------------------------------------------------------------------ R3
def conv_module(x, P, p, training, stats=None):
    """ConvModule.forward (convolution.py:226-238): PW(d->2d) -> GLU(dim=1) -> DW(k, pad (k-1)/2, no bias) -> BatchNorm1d -> Swish -> PW(d->d)."""
    s = p + ".sequential"
    d = x.shape[-1]
    u = x @ P[s + ".1.conv.weight"][:, :, 0].t() + P[s + ".1.conv.bias"]
    u = u[..., :d] * torch.sigmoid(u[..., d:])
    w = P[s + ".3.conv.weight"]                                      # [d, 1, k]
    k = w.shape[-1]
    u = F.conv1d(u.transpose(1, 2), w, None, padding=(k - 1) // 2, groups=d).transpose(1, 2)
    if training:
        mean = u.mean(dim=(0, 1))
        var = ((u - mean) ** 2).mean(dim=(0, 1))
        if stats is not None:
            n = u.shape[0] * u.shape[1]
            stats[s + ".4.running_mean"] = 0.9 * P[s + ".4.running_mean"] + 0.1 * mean.detach()
            stats[s + ".4.running_var"] = 0.9 * P[s + ".4.running_var"] + 0.1 * (var.detach() * n / max(n - 1, 1))
    else:
        mean, var = P[s + ".4.running_mean"], P[s + ".4.running_var"]
    u = (u - mean) * torch.rsqrt(var + BN_EPS) * P[s + ".4.weight"] + P[s + ".4.bias"]
    u = u * torch.sigmoid(u)
    return u @ P[s + ".6.conv.weight"][:, :, 0].t() + P[s + ".6.conv.bias"]


# ------------------------------------------------------------------------------------------------ R4
def feed_forward(x, P, p):
    """FeedForwardModule (modules.py:24-56): Linear -> Swish -> Dropout -> Linear -> Dropout."""
    s = p + ".sequential"
    h = x @ P[s + ".0.weight"].t() + P[s + ".0.bias"]
    h = h * torch.sigmoid(h)
    return h @ P[s + ".3.weight"].t() + P[s + ".3.bias"]


def _ln(x, P, p):
    return F.layer_norm(x, (x.shape[-1],), P[p + ".weight"], P[p + ".bias"], LN_EPS)


def block(x, P, p, heads, training=False, half_step_residual=False, stats=None):
    """SqueezeformerBlock.forward (encoder.py:208-247): post-LN, x = LN(x + f * sub(x)) for MHSA, FFN, Conv, FFN."""
    f = 0.5 if half_step_residual else 1.0
    s = p + ".sequential"
    x = _ln(x + rel_mhsa(x, P, s + ".0.module", heads), P, s + ".1")
    x = _ln(x + f * feed_forward(x, P, s + ".2.module"), P, s + ".3")
    x = _ln(x + conv_module(x, P, s + ".4.module", training, stats), P, s + ".5")
    x = _ln(x + f * feed_forward(x, P, s + ".6.module"), P, s + ".7")
    return x


def conv2d_subsampling(x, P):
    """DepthwiseConv2dSubsampling.forward (convolution.py:63-73): Conv2d(1->d,3,s2) -> ReLU -> depthwise Conv2d(d,3,s2) -> ReLU;
    [B,T,F] -> [B, T'', d * F''] with channel-major features."""
    y = F.relu(F.conv2d(x.unsqueeze(1), P["conv_subsample.sequential.0.weight"], P["conv_subsample.sequential.0.bias"], stride=2))
    w = P["conv_subsample.sequential.2.conv.weight"]
    y = F.relu(F.conv2d(y, w, P["conv_subsample.sequential.2.conv.bias"], stride=2, groups=w.shape[0]))
    B, C, T2, F2 = y.shape
    return y.permute(0, 2, 1, 3).contiguous().view(B, T2, C * F2)


def time_reduction(x, P):
    """TimeReductionLayer.forward (convolution.py:260-269): one 3x3 stride-2 conv over the (time, feature) plane + Swish."""
    y = F.conv2d(x.unsqueeze(1), P["time_reduction_layer.sequential.0.conv.weight"], P["time_reduction_layer.sequential.0.conv.bias"], stride=2)
    y = y * torch.sigmoid(y)
    B, C, T2, F2 = y.shape
    return y.permute(0, 2, 1, 3).contiguous().view(B, T2, C * F2)


def recover_resolution(x):
    """modules.py:137-142: every frame twice."""
    return torch.repeat_interleave(x, 2, dim=1)


def encoder(x, P, cfg: Dict, training=False, stats=None) -> Tuple[torch.Tensor, list]:
    """SqueezeformerEncoder.forward (encoder.py:135-166).  cfg: num_layers, reduce_layer_index, recover_layer_index,
    num_attention_heads, half_step_residual.  Returns (outputs, per-layer outputs); output lengths follow `>> 2 - 1`, `>> 1 - 1`,
    `* 2` (:68-69, :266-267, :162) and are computed by the caller."""
    h = conv2d_subsampling(x, P)
    h = h @ P["input_proj.0.weight"].t() + P["input_proj.0.bias"]
    taps = []
    recover_tensor = None
    for idx in range(cfg["num_layers"]):
        if idx == cfg["reduce_layer_index"]:
            recover_tensor = h
            h = time_reduction(h, P)
            h = h @ P["time_reduction_proj.weight"].t() + P["time_reduction_proj.bias"]
        if idx == cfg["recover_layer_index"]:
            h = recover_resolution(h)
            length = h.shape[1]
            h = h @ P["time_recover_layer.weight"].t() + P["time_recover_layer.bias"]
            h = h + recover_tensor[:, :length, :]
        wrapped = cfg["reduce_layer_index"] <= idx < cfg["recover_layer_index"]
        p = f"layers.{idx}" + (".module" if wrapped else "")
        y = block(h, P, p, cfg["num_attention_heads"], training, cfg.get("half_step_residual", False), stats)
        h = y + h if wrapped else y                   # ResidualConnectionModule around the middle blocks (encoder.py:88-103)
        taps.append(h)
    return h, taps


# ------------------------------------------------------------------------------------------------ parameters
def param_shapes(cfg: Dict) -> "Dict[str, tuple]":
    """state_dict keys and torch shapes, in registration order (encoder.py:72-133 and the module constructors)."""
    d, F_in, k = cfg["encoder_dim"], cfg["input_dim"], cfg["conv_kernel_size"]
    e, heads = cfg["feed_forward_expansion_factor"], cfg["num_attention_heads"]
    dh = d // heads
    F2 = ((F_in - 1) // 2 - 1) // 2
    S: Dict[str, tuple] = {}
    S["conv_subsample.sequential.0.weight"] = (d, 1, 3, 3); S["conv_subsample.sequential.0.bias"] = (d,)
    S["conv_subsample.sequential.2.conv.weight"] = (d, 1, 3, 3); S["conv_subsample.sequential.2.conv.bias"] = (d,)
    S["input_proj.0.weight"] = (d, d * F2); S["input_proj.0.bias"] = (d,)
    S["time_reduction_layer.sequential.0.conv.weight"] = (1, 1, 3, 3); S["time_reduction_layer.sequential.0.conv.bias"] = (1,)
    S["time_reduction_proj.weight"] = (d, (d - 1) // 2); S["time_reduction_proj.bias"] = (d,)
    S["time_recover_layer.weight"] = (d, d); S["time_recover_layer.bias"] = (d,)
    for idx in range(cfg["num_layers"]):
        wrapped = cfg["reduce_layer_index"] <= idx < cfg["recover_layer_index"]
        s = f"layers.{idx}" + (".module" if wrapped else "") + ".sequential"
        a = s + ".0.module.attention"
        S[a + ".u_bias"] = (heads, dh); S[a + ".v_bias"] = (heads, dh)      # nn.Parameters come before the sub-modules in a state_dict
        for n in ("query_proj", "key_proj", "value_proj"):
            S[f"{a}.{n}.weight"] = (d, d); S[f"{a}.{n}.bias"] = (d,)
        S[a + ".pos_proj.weight"] = (d, d)
        S[a + ".out_proj.weight"] = (d, d); S[a + ".out_proj.bias"] = (d,)
        S[s + ".1.weight"] = (d,); S[s + ".1.bias"] = (d,)
        for ff, ln in ((2, 3), (6, 7)):
            if ff == 6:
                c = s + ".4.module.sequential"
                S[c + ".1.conv.weight"] = (2 * d, d, 1); S[c + ".1.conv.bias"] = (2 * d,)
                S[c + ".3.conv.weight"] = (d, 1, k)
                S[c + ".4.weight"] = (d,); S[c + ".4.bias"] = (d,); S[c + ".4.running_mean"] = (d,); S[c + ".4.running_var"] = (d,)
                S[c + ".6.conv.weight"] = (d, d, 1); S[c + ".6.conv.bias"] = (d,)
                S[s + ".5.weight"] = (d,); S[s + ".5.bias"] = (d,)
            f = f"{s}.{ff}.module.sequential"
            S[f + ".0.weight"] = (d * e, d); S[f + ".0.bias"] = (d * e,)
            S[f + ".3.weight"] = (d, d * e); S[f + ".3.bias"] = (d,)
            S[f"{s}.{ln}.weight"] = (d,); S[f"{s}.{ln}.bias"] = (d,)
    return S


def init_params(cfg: Dict, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded non-trivial parameters (norm gains / biases / running statistics perturbed so that every path carries signal)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for n, s in param_shapes(cfg).items():
        if n.endswith("running_var"):
            P[n] = 0.5 + torch.rand(s, generator=g)
        elif n.endswith("running_mean"):
            P[n] = 0.2 * torch.randn(s, generator=g)
        elif len(s) == 1 and n.endswith("weight"):
            P[n] = 1.0 + 0.2 * torch.randn(s, generator=g)
        elif n.endswith("bias") and not n.endswith(("u_bias", "v_bias")):
            P[n] = 0.1 * torch.randn(s, generator=g)
        else:
            fan_in = 1
            for v in s[1:]:
                fan_in *= v
            P[n] = torch.randn(s, generator=g) * (1.0 / math.sqrt(max(fan_in, 1)))
        P[n] = P[n].to(dtype)
    return P


def output_length(T: int, cfg: Dict) -> int:
    """Frames the encoder returns for a T-frame input."""
    t = ((T - 3) // 2 + 1 - 3) // 2 + 1
    if cfg["reduce_layer_index"] < cfg["num_layers"]:
        t = (t - 3) // 2 + 1
        if cfg["recover_layer_index"] < cfg["num_layers"]:
            t = 2 * t
    return t


def squeezeformer_top(x, P, cfg: Dict, training=False, stats=None) -> torch.Tensor:
    """Squeezeformer.forward (squeezeformer/model.py:437-450): encoder -> `fc` (bias-free Linear, :431) -> log_softmax over the classes.
    P holds the top-level state_dict: `encoder.*` and `fc.weight`; cfg uses the encoder's key names (`num_layers` = num_encoder_layers)."""
    Pe = {k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}
    h, _ = encoder(x, Pe, cfg, training, stats)
    return torch.log_softmax(h @ P["fc.weight"].t(), dim=-1)
