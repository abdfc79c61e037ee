#!/usr/bin/env python3
"""Generate tests/golden/squeezeformer_top.npz by RUNNING the reference's top-level torch model — `Squeezeformer` of
/root/reference/squeezeformer/model.py:366-450 (encoder -> `fc = nn.Linear(encoder_dim, num_classes, bias=False)` -> `log_softmax`).

model.py does not import as shipped: it uses `math` (:56) and `Optional` (:77 etc.) without importing them, and its own imports go through
the package whose `__init__.py` / `modules.py` are broken (see gen_golden_squeezeformer.py).  This script reuses that script's package
plumbing (an empty package object; `squeezeformer.activation.Swish` satisfied by the reference's own Swish class) and executes model.py's
source AS IT LIES in a module namespace that already holds the two missing names (`math` = the standard module, `Optional` =
`typing.Optional`) — module plumbing only; no reference line is edited, copied or stored.  Only data is committed: seeded parameters,
the input, and the reference's outputs / autograd gradients.

Run in the build container:  python oracle/gen_golden_squeezeformer_top.py
"""
import math
import os
import sys
import types
import typing

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_squeezeformer import REF, load_reference       # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "squeezeformer_top.npz")


def load_model_py():
    load_reference()                                           # squeezeformer.{convolution,modules,...} importable
    path = os.path.join(REF, "model.py")
    mod = types.ModuleType("squeezeformer.model")
    mod.__file__ = path
    mod.__dict__.update(math=math, Optional=typing.Optional)
    exec(compile(open(path).read(), path, "exec"), mod.__dict__)
    return mod


def main():
    mod = load_model_py()
    torch.manual_seed(20250302)
    cfg = dict(num_classes=29, input_dim=20, encoder_dim=32, num_encoder_layers=4, reduce_layer_index=1, recover_layer_index=3,
               num_attention_heads=4, feed_forward_expansion_factor=4, conv_expansion_factor=2, conv_kernel_size=7, half_step_residual=True)
    model = mod.Squeezeformer(input_dropout_p=0.0, feed_forward_dropout_p=0.0, attention_dropout_p=0.0, conv_dropout_p=0.0, **cfg)
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and n.endswith("weight"): p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias") and not n.endswith(("u_bias", "v_bias")): p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in model.named_buffers():
            if n.endswith("running_mean"): b.copy_(0.2 * torch.randn(b.shape, generator=g))
            if n.endswith("running_var"): b.copy_(0.5 + torch.rand(b.shape, generator=g))
    out = {"cfg_keys": np.array(list(cfg)), "cfg_vals": np.array([int(v) for v in cfg.values()], dtype=np.int64)}
    for k, v in model.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            out["sd/" + k] = v.numpy().copy()
    B, T = 2, 90
    x = torch.randn(B, T, cfg["input_dim"], generator=g)
    lengths = torch.tensor([T, T - 9])
    out["x"], out["lengths"] = x.numpy(), lengths.numpy()
    model.eval()
    with torch.no_grad():
        y, yl = model(x, lengths.clone())
    out["eval_y"], out["eval_len"] = y.numpy(), yl.numpy()
    # (model.count_parameters() raises in the reference: model.py:249 sums `p.numel` without calling it)
    model.train()
    xg = x.clone().requires_grad_(True)
    y, _ = model(xg, lengths.clone())
    G = torch.randn(y.shape, generator=g)
    (y * G).sum().backward()
    out["train_G"], out["train_y"], out["train_dx"] = G.numpy(), y.detach().numpy(), xg.grad.numpy()
    for n, p in model.named_parameters():
        out["train_grad/" + n] = p.grad.numpy().copy()
    # a CTC loss on the reference's log-probabilities (how a caller would train this model), with torch's own ctc_loss
    yc, ylc = model(x, lengths.clone())
    tgt = torch.randint(1, cfg["num_classes"], (B, 5), generator=g)
    loss = torch.nn.functional.ctc_loss(yc.transpose(0, 1), tgt, ylc, torch.tensor([5, 4]), blank=0, reduction="sum")
    out["ctc_targets"], out["ctc_loss"] = tgt.numpy(), np.array(float(loss), dtype=np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays; y", tuple(y.shape), "lengths", yl.tolist(), "ctc", float(loss))


if __name__ == "__main__":
    sys.exit(main())
