"""Host-side mirror of the reference's torch Squeezeformer encoder — `squeezeformer/encoder.py:26-166` (`SqueezeformerEncoder`)
with its relative-position attention (`attention.py:25-139`), convolution module (`convolution.py:199-238`), conv2d subsampling
(`:39-73`), time reduction (`:241-269`) and recovery (`modules.py:137-142`) — on the HIP library (csrc/squeezeformer_r4.hip,
family ISHARA_FAMILY_TORCH_SQUEEZEFORMER).

`SqueezeformerEncoder(input_dim, encoder_dim, num_layers, reduce_layer_index, recover_layer_index, num_attention_heads, ...)` has
the reference's constructor; `enc(inputs, input_lengths) -> (outputs, output_lengths)` its forward contract; `state_dict()` /
`load_state_dict()` use the reference's keys and torch layouts; the call is differentiable through torch.autograd exactly like
`ishara_amd.ConformerEncoder`.  The device buffers are planned for one clip length: `seq_len` frames per input clip.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import _lib
from .conformer import _TorchFamilyEncoder


class _SqueezeformerLayout:
    """torch state_dict array <-> the library's array, by key."""

    def to_lib(self, name: str, a: np.ndarray) -> np.ndarray:
        if a.ndim == 4: return np.ascontiguousarray(a.reshape(a.shape[0], 9)) if a.shape[0] > 1 else np.ascontiguousarray(a.reshape(9))     # 3x3 kernels
        if a.ndim == 3 and a.shape[1] == 1: return np.ascontiguousarray(a[:, 0, :].T)         # depthwise [d,1,k] -> [k,d]
        if a.ndim == 3: return np.ascontiguousarray(a[:, :, 0].T)                             # pointwise [out,in,1] -> [in,out]
        if name.endswith(("u_bias", "v_bias")): return np.ascontiguousarray(a.reshape(-1))    # [H,dh] -> [d]
        if a.ndim == 2: return np.ascontiguousarray(a.T)                                      # Linear [out,in] -> [in,out]
        return a

    def to_torch(self, name: str, a: np.ndarray, heads: int = 1) -> np.ndarray:
        if name.endswith("conv_subsample.sequential.0.weight") or name.endswith("conv_subsample.sequential.2.conv.weight"): return np.ascontiguousarray(a.reshape(a.shape[0], 1, 3, 3))
        if name.endswith("time_reduction_layer.sequential.0.conv.weight"): return np.ascontiguousarray(a.reshape(1, 1, 3, 3))
        if name.endswith(".3.conv.weight"): return np.ascontiguousarray(a.T[:, None, :])
        if name.endswith((".1.conv.weight", ".6.conv.weight")): return np.ascontiguousarray(a.T[:, :, None])
        if name.endswith(("u_bias", "v_bias")): return np.ascontiguousarray(a.reshape(heads, -1))
        if a.ndim == 2: return np.ascontiguousarray(a.T)
        return a


class SqueezeformerEncoder(_TorchFamilyEncoder):
    def __init__(self, input_dim: int = 80, encoder_dim: int = 512, num_layers: int = 16, reduce_layer_index: int = 7, recover_layer_index: int = 15,
                 num_attention_heads: int = 8, feed_forward_expansion_factor: int = 4, conv_expansion_factor: int = 2, input_dropout_p: float = 0.1,
                 feed_forward_dropout_p: float = 0.1, attention_dropout_p: float = 0.1, conv_dropout_p: float = 0.1, conv_kernel_size: int = 31,
                 half_step_residual: bool = False, *, seq_len: int = 384, max_batch: int = 16, dtype: str = "bf16", device: Optional[str] = "cuda:0", seed: int = 0):
        if conv_expansion_factor != 2:
            raise ValueError("Currently, Only Supports expansion_factor 2")             # convolution.py:222
        ps = {input_dropout_p, feed_forward_dropout_p, attention_dropout_p, conv_dropout_p}
        if len(ps) != 1:
            raise ValueError("the library takes one dropout probability for the five dropout sites; pass equal values")
        cfg = _lib.Config()
        cfg.family = _lib.FAMILY_TORCH_SQUEEZEFORMER
        cfg.dim, cfg.num_conv_conform_blocks, cfg.num_heads = encoder_dim, num_layers, num_attention_heads
        cfg.expansion_factor, cfg.transformer_kernel_size, cfg.dropout_rate = feed_forward_expansion_factor, conv_kernel_size, float(ps.pop())
        cfg.frames, cfg.features, cfg.num_classes = seq_len, input_dim, 60
        cfg.dtype = {"f32": _lib.F32, "bf16": _lib.BF16}[dtype]
        cfg.max_batch, cfg.max_label_len, cfg.attn_impl = max_batch, 64, 1
        cfg.reduce_layer_index, cfg.recover_layer_index, cfg.half_step_residual = reduce_layer_index, recover_layer_index, int(half_step_residual)
        self.dim, self.num_layers, self.num_heads = encoder_dim, num_layers, num_attention_heads
        self.reduce_layer_index, self.recover_layer_index = reduce_layer_index, recover_layer_index
        lay = _SqueezeformerLayout()
        heads = num_attention_heads
        base_to_torch = lay.to_torch
        lay.to_torch = lambda name, a: base_to_torch(name, a, heads)
        self._create(cfg, lay, input_dim, seq_len, max_batch, device, seed)

    def _default_state(self, seed):
        """nn.Linear / nn.Conv default initialisers (U(+-1/sqrt(fan_in)) for weight and bias), xavier_uniform u/v biases, unit norms,
        (0, 1) running statistics: the distributions torch would draw from, not its draws."""
        g = torch.Generator().manual_seed(seed)
        sd = OrderedDict()

        def uniform(shape, lim):
            return ((torch.rand(tuple(shape), generator=g) * 2 - 1) * lim).numpy().astype(np.float32)

        shapes = self.torch_shapes()
        for name, shape in shapes.items():
            if name.endswith("running_mean"): sd[name] = np.zeros(shape, np.float32)
            elif name.endswith("running_var"): sd[name] = np.ones(shape, np.float32)
            elif name.endswith(("u_bias", "v_bias")): sd[name] = uniform(shape, (6.0 / (shape[0] + shape[1])) ** 0.5)
            elif len(shape) == 1 and name.endswith("weight"): sd[name] = np.ones(shape, np.float32)                     # LayerNorm / BatchNorm gains
            elif name.endswith("weight"): sd[name] = uniform(shape, float(np.prod(shape[1:])) ** -0.5)
            else:
                w = sd.get(name[:-4] + "weight")
                sd[name] = np.zeros(shape, np.float32) if w is None or w.ndim == 1 else uniform(shape, float(np.prod(w.shape[1:])) ** -0.5)
        return sd

    def output_lengths(self, input_lengths: torch.Tensor) -> torch.Tensor:
        """`>> 2, -1` (convolution.py:68-69), `>> 1, -1` at the reduction layer (:266-267), `* 2` at the recovery layer (encoder.py:162)."""
        out = (input_lengths >> 2) - 1
        if self.reduce_layer_index < self.num_layers:
            out = (out >> 1) - 1
            if self.recover_layer_index < self.num_layers:
                out = out * 2
        return out

    def __call__(self, inputs, input_lengths=None):
        """SqueezeformerEncoder.forward(inputs, input_lengths) — encoder.py:135-166: (outputs [B, T_out, encoder_dim], output_lengths)."""
        y = self._apply(inputs)
        if input_lengths is None:
            return y
        return y, self.output_lengths(torch.as_tensor(input_lengths).clone())

    forward = __call__
