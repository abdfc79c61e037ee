#!/usr/bin/env python3
"""Generate tests/golden/conformer_r5.npz by RUNNING THE REFERENCE: imports
/root/reference/conformer/conformer.py (SURVEY §8a row R5 — the only hot-path file of the reference
that imports and runs here), builds a seeded ConformerEncoder in eval mode, and stores its
state_dict, a seeded input and the reference outputs (per block and per sub-module of block 0).

Only data is committed (inputs / weights / expected outputs); the reference source is not copied.
Run in the build container:  python oracle/gen_golden_conformer.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/conformer/conformer.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "conformer_r5.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_conformer", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.manual_seed(20250227)
    dim, layers, heads, ksize, exp = 32, 2, 4, 15, 4
    enc = ref.ConformerEncoder(dim=dim, num_layers=layers, num_heads=heads, expansion_factor=exp, kernel_size=ksize, dropout=0.1)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():       # make every parameter path non-trivial (defaults have unit norms / zero running stats)
        for n, p in enc.named_parameters():
            if n.endswith("layer_norm.weight") or n.endswith("batch_norm.weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in enc.named_buffers():
            if n.endswith("running_mean"): b.copy_(0.2 * torch.randn(b.shape, generator=g))
            if n.endswith("running_var"): b.copy_(0.5 + torch.rand(b.shape, generator=g))
    enc.eval()
    x = torch.randn(2, 48, dim, generator=g)
    out = {"x": x.numpy(), "cfg": np.array([dim, layers, heads, ksize, exp], dtype=np.int64)}
    with torch.no_grad():
        h = x
        for i, layer in enumerate(enc.layers):
            if i == 0:
                a = layer.ffn1(h); out["blk0_ffn1"] = a.numpy()
                b = layer.attention(a); out["blk0_attn"] = b.numpy()
                c = layer.conv(b); out["blk0_conv"] = c.numpy()
                d = layer.ffn2(c); out["blk0_ffn2"] = d.numpy()
            h = layer(h)
            out[f"blk{i}_out"] = h.numpy()
        assert torch.allclose(h, enc(x))
    # a training-mode gradient pin for one block (dropout off so it is deterministic)
    enc2 = ref.ConformerEncoder(dim=dim, num_layers=1, num_heads=heads, expansion_factor=exp, kernel_size=ksize, dropout=0.0)
    enc2.load_state_dict({k: v for k, v in enc.state_dict().items() if k.startswith("layers.0.")}, strict=True)
    enc2.train()
    xg = x.clone().requires_grad_(True)
    G = torch.randn(2, 48, dim, generator=g)
    y = enc2(xg)
    (y * G).sum().backward()
    out["train_G"] = G.numpy(); out["train_y"] = y.detach().numpy(); out["train_dx"] = xg.grad.numpy()
    for n, p in enc2.named_parameters():
        out["train_grad/" + n] = p.grad.numpy()
    for k, v in enc.state_dict().items():
        out["sd/" + k] = v.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
