"""CPU restatement of the reference's torch Conformer stack:
/root/reference/conformer/conformer.py:6-87 — post-LN FeedForwardModule (:6-22), MultiHeadSelfAttention on
nn.MultiheadAttention (:24-35; in_proj [3d,d] packed q|k|v block-major, scale dh**-0.5, biases), ConvolutionModule
(:37-57; pointwise -> GLU(dim=1) -> depthwise k, pad k//2, bias -> BatchNorm1d -> pointwise; no activation, dropout
unused), ConformerBlock with a trailing LayerNorm (:59-73), ConformerEncoder (:76-87).

TEST INFRASTRUCTURE ONLY.  PARITY PINNED: tests/test_golden_conformer.py checks this restatement against golden
vectors produced by running the reference file itself (oracle/gen_golden_conformer.py -> tests/golden/conformer_r5.npz).
State-dict keys are the reference's (`layers.{i}.ffn1.linear1.weight`, ...).  Pure functional torch; dropout = 0.
"""
import torch
import torch.nn.functional as F

LN_EPS = 1e-5      # nn.LayerNorm default
BN_EPS = 1e-5      # nn.BatchNorm1d default (momentum 0.1)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def ffn(x, sd, p):
    h = F.silu(x @ sd[p + ".linear1.weight"].t() + sd[p + ".linear1.bias"])
    h = h @ sd[p + ".linear2.weight"].t() + sd[p + ".linear2.bias"]
    return _ln(h + x, sd, p + ".layer_norm")


def mhsa(x, sd, p, heads):
    B, T, d = x.shape
    dh = d // heads
    qkv = x @ sd[p + ".attention.in_proj_weight"].t() + sd[p + ".attention.in_proj_bias"]
    q, k, v = [t.view(B, T, heads, dh).permute(0, 2, 1, 3) for t in qkv.split(d, dim=-1)]
    a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    o = o @ sd[p + ".attention.out_proj.weight"].t() + sd[p + ".attention.out_proj.bias"]
    return _ln(o + x, sd, p + ".layer_norm")


def conv_module(x, sd, p, training=False):
    d = x.shape[-1]
    u = x @ sd[p + ".pointwise_conv1.weight"][:, :, 0].t() + sd[p + ".pointwise_conv1.bias"]
    u = u[..., :d] * torch.sigmoid(u[..., d:])                                   # nn.GLU(dim=1) on [B, 2d, T]
    w = sd[p + ".depthwise_conv.weight"]                                          # [d, 1, k]
    k = w.shape[-1]
    u = F.conv1d(u.transpose(1, 2), w, sd[p + ".depthwise_conv.bias"], padding=k // 2, groups=d).transpose(1, 2)
    if training:
        mean = u.mean(dim=(0, 1)); var = ((u - mean) ** 2).mean(dim=(0, 1))       # biased batch variance
    else:
        mean, var = sd[p + ".batch_norm.running_mean"], sd[p + ".batch_norm.running_var"]
    u = (u - mean) * torch.rsqrt(var + BN_EPS) * sd[p + ".batch_norm.weight"] + sd[p + ".batch_norm.bias"]
    u = u @ sd[p + ".pointwise_conv2.weight"][:, :, 0].t() + sd[p + ".pointwise_conv2.bias"]
    return _ln(u + x, sd, p + ".layer_norm")


def block(x, sd, p, heads, training=False, taps=None):
    a = ffn(x, sd, p + ".ffn1")
    b = mhsa(a, sd, p + ".attention", heads)
    c = conv_module(b, sd, p + ".conv", training)
    e = ffn(c, sd, p + ".ffn2")
    if taps is not None:
        taps.update(ffn1=a, attn=b, conv=c, ffn2=e)
    return _ln(e, sd, p + ".layer_norm")


def encoder(x, sd, num_layers, heads, training=False):
    outs = []
    for i in range(num_layers):
        x = block(x, sd, f"layers.{i}", heads, training)
        outs.append(x)
    return x, outs
