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


class _FcLogSoftmaxFn(torch.autograd.Function):
    """log_softmax(x @ W^T) over the class axis and its backward, every product and reduction in libishara_hip.so (`ishara_op_dense_fwd_ex`
    / `ishara_op_dense_bwd` in exact-fp32 MFMA mode — the reference's output layer is fp32 — and `ishara_op_log_softmax_fwd / _bwd`);
    torch only owns the buffers."""

    @staticmethod
    def forward(ctx, x, weight, lib):
        import ctypes as C
        from .model import _stream
        B, T, d = x.shape
        M, N = B * T, weight.shape[0]
        Np = (N + 7) // 8 * 8                                      # class count padded to the GEMMs' 16-byte row alignment (zero weight columns)
        x2 = x.detach().reshape(M, d).contiguous()
        Wk = torch.zeros(d, Np, dtype=torch.float32, device=x.device)
        Wk[:, :N] = weight.detach().t()                            # nn.Linear [C, d] -> the library's [K, N]
        z = torch.empty(M, Np, dtype=torch.float32, device=x.device)
        y = torch.empty_like(z)
        sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, d, Np)) + 256, dtype=torch.uint8, device=x.device)
        scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
        _lib.check(lib.ishara_op_dense_fwd_ex(_lib.F32, _lib.ptr(x2), _lib.ptr(Wk), None, None, _lib.ptr(z), M, d, Np, 0, scp, _stream()), "fc")
        _lib.check(lib.ishara_op_log_softmax_fwd(_lib.ptr(z), _lib.ptr(y), M, N, Np, _stream()), "log_softmax")
        ctx.save_for_backward(x2, Wk, y)
        ctx.lib, ctx.scratch, ctx.shape = lib, (sc, scp), (B, T, d, N, Np)
        return y[:, :N].reshape(B, T, N)

    @staticmethod
    def backward(ctx, dy):
        from .model import _stream
        x2, Wk, y = ctx.saved_tensors
        lib, (sc, scp), (B, T, d, N, Np) = ctx.lib, ctx.scratch, ctx.shape
        M = B * T
        dy2 = torch.zeros(M, Np, dtype=torch.float32, device=dy.device)
        dy2[:, :N] = dy.detach().reshape(M, N)
        dz = torch.empty_like(dy2)
        _lib.check(lib.ishara_op_log_softmax_bwd(_lib.ptr(dy2), _lib.ptr(y), _lib.ptr(dz), M, N, Np, _stream()), "log_softmax_bwd")
        dx = torch.empty(M, d, dtype=torch.float32, device=dy.device)
        dW = torch.zeros(d, Np, dtype=torch.float32, device=dy.device)
        db = torch.zeros(Np, dtype=torch.float32, device=dy.device)
        _lib.check(lib.ishara_op_dense_bwd(_lib.F32, _lib.ptr(x2), _lib.ptr(Wk), _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), M, d, Np, scp, _stream()), "fc_bwd")
        return dx.reshape(B, T, d), dW[:, :N].t().contiguous(), None


class Squeezeformer:
    """`Squeezeformer(num_classes, input_dim, encoder_dim, ...)` of squeezeformer/model.py:366-450: the encoder above, the bias-free output
    layer `fc = nn.Linear(encoder_dim, num_classes, bias=False)` (:431) and `log_softmax` (:449).  `model(inputs, input_lengths)` returns
    `(log_probs [B, T_out, num_classes], output_lengths)` like the reference's forward (:437-450) and is differentiable end to end (the
    encoder through `ishara_encoder_backward`, the head through the library's dense / log-softmax operators), so that
    `torch.nn.functional.ctc_loss(log_probs.transpose(0, 1), ...)` or the library's own CTC kernel train it.  state_dict keys:
    `encoder.<the encoder's keys>` and `fc.weight` [num_classes, encoder_dim]."""

    def __init__(self, num_classes: int, input_dim: int = 80, encoder_dim: int = 512, num_encoder_layers: int = 16, reduce_layer_index: int = 7,
                 recover_layer_index: int = 15, num_attention_heads: int = 8, feed_forward_expansion_factor: int = 4, conv_expansion_factor: int = 2,
                 input_dropout_p: float = 0.1, feed_forward_dropout_p: float = 0.1, attention_dropout_p: float = 0.1, conv_dropout_p: float = 0.1,
                 conv_kernel_size: int = 31, half_step_residual: bool = False, *, seq_len: int = 384, max_batch: int = 16, dtype: str = "bf16",
                 device: Optional[str] = "cuda:0", seed: int = 0):
        self.encoder = SqueezeformerEncoder(input_dim, encoder_dim, num_encoder_layers, reduce_layer_index, recover_layer_index, num_attention_heads,
                                            feed_forward_expansion_factor, conv_expansion_factor, input_dropout_p, feed_forward_dropout_p,
                                            attention_dropout_p, conv_dropout_p, conv_kernel_size, half_step_residual,
                                            seq_len=seq_len, max_batch=max_batch, dtype=dtype, device=device, seed=seed)
        self.num_classes = num_classes
        g = torch.Generator().manual_seed(seed + 1)
        w = (torch.rand(num_classes, encoder_dim, generator=g) * 2 - 1) * encoder_dim ** -0.5        # nn.Linear's default U(+-1/sqrt(fan_in))
        self.fc_weight = torch.nn.Parameter(w.to(self.encoder.device), requires_grad=True)
        self.training = True

    def train(self, mode: bool = True):
        self.training = bool(mode)
        self.encoder.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return self.encoder.parameters() + [self.fc_weight]

    def zero_grad(self):
        self.encoder.zero_grad()
        self.fc_weight.grad = None

    def count_parameters(self) -> int:
        """model.py:433-435 means the ENCODER's parameter count (the reference's own method raises: :249 sums `p.numel` without calling it)."""
        return self.encoder.n_train

    def state_dict(self):
        sd = OrderedDict(("encoder." + k, v) for k, v in self.encoder.state_dict().items())
        sd["fc.weight"] = self.fc_weight.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd, strict: bool = True):
        self.encoder.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=strict)
        if "fc.weight" in sd:
            with torch.no_grad():
                self.fc_weight.copy_(torch.as_tensor(sd["fc.weight"]).to(self.fc_weight.device, torch.float32))
        elif strict:
            raise KeyError("load_state_dict: missing fc.weight")

    def __call__(self, inputs, input_lengths):
        enc_out, out_len = self.encoder(inputs, input_lengths)
        if torch.is_grad_enabled() and self.training:
            return _FcLogSoftmaxFn.apply(enc_out, self.fc_weight, self.encoder._lib), out_len
        with torch.no_grad():
            return _FcLogSoftmaxFn.apply(enc_out, self.fc_weight, self.encoder._lib), out_len

    forward = __call__
