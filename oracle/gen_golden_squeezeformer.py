#!/usr/bin/env python3
"""Generate tests/golden/squeezeformer_r4.npz by RUNNING THE REFERENCE's torch Squeezeformer files:
/root/reference/squeezeformer/{attention,modules,convolution,encoder}.py.

The package does not import as shipped: `squeezeformer/__init__.py:15` pulls in model.py (missing imports) and `modules.py:21`
imports `squeezeformer.activation`, a module the repository lacks.  This script therefore (a) registers an empty package object
for `squeezeformer` whose `__path__` is the reference directory, so the four files load WITHOUT executing `__init__.py`, and
(b) satisfies `squeezeformer.activation.Swish` with the `Swish` class the reference itself defines in convolution.py:22-27 —
the class statement is taken from that file's syntax tree and executed as is; no line of the reference is edited or copied into
this repository.  Only data is committed: seeded parameters, inputs, and the reference's outputs / autograd gradients.

Run in the build container:  python oracle/gen_golden_squeezeformer.py
"""
import ast
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/squeezeformer"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "squeezeformer_r4.npz")


def load_reference():
    pkg = types.ModuleType("squeezeformer")
    pkg.__path__ = [REF]
    sys.modules["squeezeformer"] = pkg
    tree = ast.parse(open(os.path.join(REF, "convolution.py")).read())
    swish = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Swish"]
    ns = {"nn": torch.nn, "Tensor": torch.Tensor}
    exec(compile(ast.Module(body=swish, type_ignores=[]), os.path.join(REF, "convolution.py"), "exec"), ns)
    act = types.ModuleType("squeezeformer.activation")
    act.Swish = ns["Swish"]
    sys.modules["squeezeformer.activation"] = act
    return importlib.import_module("squeezeformer.encoder"), importlib.import_module("squeezeformer.attention"), importlib.import_module("squeezeformer.modules")


def main():
    enc_mod, att_mod, mod_mod = load_reference()
    torch.manual_seed(20250301)
    cfg = dict(input_dim=20, encoder_dim=32, num_layers=4, reduce_layer_index=1, recover_layer_index=3, num_attention_heads=4,
               feed_forward_expansion_factor=4, conv_expansion_factor=2, conv_kernel_size=7, half_step_residual=True)
    enc = enc_mod.SqueezeformerEncoder(input_dropout_p=0.0, feed_forward_dropout_p=0.0, attention_dropout_p=0.0, conv_dropout_p=0.0, **cfg)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():       # make every parameter path non-trivial
        for n, p in enc.named_parameters():
            if p.dim() == 1 and n.endswith("weight"): p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias") and not n.endswith(("u_bias", "v_bias")): p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in enc.named_buffers():
            if n.endswith("running_mean"): b.copy_(0.2 * torch.randn(b.shape, generator=g))
            if n.endswith("running_var"): b.copy_(0.5 + torch.rand(b.shape, generator=g))
    out = {"cfg_keys": np.array(list(cfg)), "cfg_vals": np.array([int(v) for v in cfg.values()], dtype=np.int64)}
    for k, v in enc.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            out["sd/" + k] = v.numpy().copy()
    B, T = 2, 90
    x = torch.randn(B, T, cfg["input_dim"], generator=g)
    lengths = torch.tensor([T, T - 7])
    out["x"] = x.numpy()
    # ---- R2: the table; R1: one attention module alone (with and without the surrounding residual / LN)
    out["pe_T13"] = mod_mod.RelPositionalEncoding(cfg["encoder_dim"])(torch.zeros(1, 13, cfg["encoder_dim"])).numpy()
    enc.eval()
    with torch.no_grad():
        h, hl = enc.conv_subsample(x, lengths.clone())
        out["subsample"] = h.numpy(); out["subsample_len"] = hl.numpy()
        h = enc.input_proj(h)
        out["input_proj"] = h.numpy()
        blk0 = enc.layers[0]
        out["blk0_mhsa"] = blk0.sequential[0].module(h).numpy()                 # MultiHeadedSelfAttentionModule alone (R1)
        a = blk0.sequential[1](blk0.sequential[0](h)); out["blk0_after_attn_ln"] = a.numpy()
        b = blk0.sequential[3](blk0.sequential[2](a)); out["blk0_after_ffn1_ln"] = b.numpy()
        out["blk0_conv"] = blk0.sequential[4].module(b).numpy()                  # ConvModule alone, eval mode (R3)
        y, yl = enc(x, lengths.clone())
        out["eval_y"] = y.numpy(); out["eval_len"] = yl.numpy()
        # per-layer outputs through a second pass with hooks
        taps = []
        hooks = [l.register_forward_hook(lambda m, i, o: taps.append(o.detach().numpy().copy())) for l in enc.layers]
        enc(x, lengths.clone())
        for hk in hooks: hk.remove()
        for i, t in enumerate(taps): out[f"eval_layer{i}"] = t
    # ---- training mode (BatchNorm batch statistics), autograd gradients of sum(y * G)
    enc.train()
    xg = x.clone().requires_grad_(True)
    y, _ = enc(xg, lengths.clone())
    G = torch.randn(y.shape, generator=g)
    (y * G).sum().backward()
    out["train_G"] = G.numpy(); out["train_y"] = y.detach().numpy(); out["train_dx"] = xg.grad.numpy()
    for n, p in enc.named_parameters():
        out["train_grad/" + n] = p.grad.numpy().copy()
    for n, b in enc.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            out["train_stats/" + n] = b.numpy().copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays; y", tuple(y.shape))


if __name__ == "__main__":
    sys.exit(main())
