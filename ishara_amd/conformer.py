"""Host-side mirror of the reference's torch Conformer stack — `conformer/conformer.py:76-87` (`ConformerEncoder`) and the
modules it stacks (`:6-73`) — on the HIP library (csrc/conformer_r5.hip, family ISHARA_FAMILY_TORCH_CONFORMER).

`ConformerEncoder(dim, num_layers=7, num_heads=8, expansion_factor=4, kernel_size=31, dropout=0.1)` has the reference's
constructor; `enc(x)` / `enc.train()` / `enc.eval()` / `state_dict()` / `load_state_dict()` behave like the torch module's:
state-dict keys and array layouts are the reference's (`layers.0.ffn1.linear1.weight` `[out,in]`, `in_proj_weight` `[3d,d]`
q|k|v block-major, `depthwise_conv.weight` `[d,1,k]`, `batch_norm.running_mean`, ...).  `enc(x)` is differentiable through
`torch.autograd`: `y = enc(x); loss(y).backward()` fills `x.grad` and `enc.flat.grad` (the flat trainable-parameter vector
any `torch.optim` optimiser can step); `enc.grad_state_dict()` gives the parameter gradients keyed and laid out like
`state_dict()`.

All arithmetic runs in libishara_hip.so; torch tensors are containers.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import IsharaError


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def head_major_perm(dim: int, heads: int) -> np.ndarray:
    """Column order of the library's packed qkv projection (h*3dh + {q,k,v}*dh + i) in terms of nn.MultiheadAttention's
    in_proj rows ({q,k,v}*d + h*dh + i): `W_lib[:, j] = in_proj_weight[perm[j], :]`."""
    dh = dim // heads
    return np.array([part * dim + h * dh + i for h in range(heads) for part in range(3) for i in range(dh)], dtype=np.int64)


class _LayoutMap:
    """torch state_dict array <-> the array stored in the library's flat buffer, per entry name."""

    def __init__(self, dim: int, heads: int):
        self.perm = head_major_perm(dim, heads)
        self.inv = np.argsort(self.perm)

    def to_lib(self, name: str, a: np.ndarray) -> np.ndarray:
        if name.endswith("in_proj_weight"): return np.ascontiguousarray(a[self.perm].T)            # [3d,d] -> [d,3d] head-major columns
        if name.endswith("in_proj_bias"): return np.ascontiguousarray(a[self.perm])
        if name.endswith("depthwise_conv.weight"): return np.ascontiguousarray(a[:, 0, :].T)        # [d,1,k] -> [k,d]
        if name.endswith("pointwise_conv1.weight") or name.endswith("pointwise_conv2.weight"): return np.ascontiguousarray(a[:, :, 0].T)
        if a.ndim == 2: return np.ascontiguousarray(a.T)                                            # Linear [out,in] -> [in,out]
        return a

    def to_torch(self, name: str, a: np.ndarray) -> np.ndarray:
        if name.endswith("in_proj_weight"): return np.ascontiguousarray(a.T[self.inv])
        if name.endswith("in_proj_bias"): return np.ascontiguousarray(a[self.inv])
        if name.endswith("depthwise_conv.weight"): return np.ascontiguousarray(a.T[:, None, :])
        if name.endswith("pointwise_conv1.weight") or name.endswith("pointwise_conv2.weight"): return np.ascontiguousarray(a.T[:, :, None])
        if a.ndim == 2: return np.ascontiguousarray(a.T)
        return a


def default_state_dict(shapes: "OrderedDict[str, tuple]", seed: int) -> "OrderedDict[str, np.ndarray]":
    """Default initialisers of the torch layers the reference stacks: nn.Linear / nn.Conv1d draw weight and bias from
    U(+-1/sqrt(fan_in)); nn.MultiheadAttention uses xavier_uniform for in_proj_weight and zero in_proj / out_proj biases;
    norms start at (1, 0), running statistics at (0, 1).  Same distributions as torch's, not the same draws."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def uniform(shape, lim):
        return ((torch.rand(tuple(shape), generator=g) * 2 - 1) * lim).numpy().astype(np.float32)

    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if "layer_norm" in name or "batch_norm" in name:
            sd[name] = (np.ones if leaf in ("weight", "running_var") else np.zeros)(shape, np.float32)
        elif name.endswith("in_proj_weight"):
            sd[name] = uniform(shape, (6.0 / (shape[0] + shape[1])) ** 0.5)
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            sd[name] = np.zeros(shape, np.float32)
        elif leaf == "weight":
            sd[name] = uniform(shape, float(np.prod(shape[1:])) ** -0.5)
        else:                                       # bias of a Linear / Conv1d: the fan_in of its weight
            sd[name] = uniform(shape, float(np.prod(sd[name[:-4] + "weight"].shape[1:])) ** -0.5)
    return sd


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, enc):
        ctx.enc = enc
        y = enc._forward(x, training=enc.training)
        ctx.gen = enc._gen                # the library keeps the saved activations of its LAST training forward only
        return y

    @staticmethod
    def backward(ctx, dy):
        enc = ctx.enc
        if ctx.gen != enc._gen:
            raise IsharaError("backward() of an encoder output whose saved activations were overwritten by a later forward pass of the same "
                              "encoder: one live autograd graph per encoder (call backward() before the next forward, or use a second encoder "
                              "object for a second branch)")
        dx = enc._backward(dy)
        return dx, enc.grads[:enc.n_train].clone(), None


class _TorchFamilyEncoder:
    """Shared host side of the two torch encoder families: one library handle, flat parameter / gradient buffers, state_dict in
    the reference's keys and layouts, torch.autograd integration.  Subclasses fill an `_lib.Config` and a layout map."""

    def _create(self, cfg, layout_map, in_features, seq_len, max_batch, device, seed):
        self._lib = _lib.load()
        self._cfg = cfg
        self.T, self.F_in, self.max_batch = seq_len, in_features, max_batch
        self._map = layout_map
        self._h = C.c_void_p()
        _lib.check(self._lib.ishara_create(C.byref(cfg), C.byref(self._h)), "ishara_create")
        self.n_total = int(self._lib.ishara_param_total(self._h))
        self.n_train = int(self._lib.ishara_param_trainable(self._h))
        self.entries = []
        for i in range(self._lib.ishara_param_entries(self._h)):
            name, nd, sh, off, tr = C.c_char_p(), C.c_int32(), (C.c_int64 * 2)(), C.c_int64(), C.c_int32()
            _lib.check(self._lib.ishara_param_info(self._h, i, C.byref(name), C.byref(nd), C.byref(sh), C.byref(off), C.byref(tr)))
            shape = (int(sh[0]),) if nd.value == 1 else (int(sh[0]), int(sh[1]))
            self.entries.append((name.value.decode(), shape, int(off.value), bool(tr.value)))
        self.T_out = int(self._lib.ishara_encoder_output_frames(self._h))
        self.training = True
        self.device = None
        self._seed, self._steps = seed * 7919 + 17, 0
        self._gen = 0                     # forward passes so far (autograd graphs check it: the library saves ONE pass)
        self._synced_version = -1         # `flat._version` at the last ishara_sync_weights
        if device is not None:
            self._to_device(device, seed)

    # ------------------------------------------------------------------ device state
    def _to_device(self, device, seed):
        if not torch.cuda.is_available():
            raise IsharaError("ishara_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.device = dev = torch.device(device)
        torch.cuda.set_device(dev)
        self.params = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.opt_m = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.opt_v = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.opt_slow = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        wsb = int(self._lib.ishara_workspace_bytes(self._h))
        self.workspace = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
        ws_ptr = self.workspace.data_ptr() + (-self.workspace.data_ptr()) % 256
        _lib.check(self._lib.ishara_bind(self._h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.opt_m), _lib.ptr(self.opt_v),
                                         _lib.ptr(self.opt_slow), C.c_void_p(ws_ptr), wsb), "ishara_bind")
        # the flat trainable vector as a leaf torch optimisers can step; it aliases the library's parameter buffer
        self.flat = torch.nn.Parameter(self.params[:self.n_train], requires_grad=True)
        self.load_state_dict(self._default_state(seed))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ishara_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ nn.Module surface
    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [self.flat]

    def zero_grad(self):
        self.flat.grad = None

    def torch_shapes(self) -> "OrderedDict[str, tuple]":
        """state_dict key -> shape in the reference's layout, in state_dict order."""
        out = OrderedDict()
        for n, s, _, _ in self.entries:
            out[n] = tuple(self._map.to_torch(n, np.empty(s, np.float32)).shape)
        return out

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        flat = self.params.detach().cpu().numpy()
        return OrderedDict((n, torch.from_numpy(self._map.to_torch(n, flat[o:o + int(np.prod(s))].reshape(s).copy()))) for n, s, o, _ in self.entries)

    def grad_state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Parameter gradients of the last backward pass, keyed and laid out like state_dict()."""
        flat = self.grads.detach().cpu().numpy()
        return OrderedDict((n, torch.from_numpy(self._map.to_torch(n, flat[o:o + int(np.prod(s))].reshape(s).copy())))
                           for n, s, o, t in self.entries if t)

    def load_state_dict(self, sd: Dict[str, "torch.Tensor"], strict: bool = True):
        names = {n for n, _, _, _ in self.entries}
        extra = [k for k in sd if k not in names and not k.endswith("num_batches_tracked")]
        missing = [n for n in names if n not in sd]
        if strict and (extra or missing):
            raise KeyError(f"load_state_dict: missing {sorted(missing)[:4]}, unexpected {sorted(extra)[:4]}")
        flat = self.params.detach().cpu().numpy().copy()
        for n, s, o, _ in self.entries:
            if n not in sd:
                continue
            a = np.asarray(sd[n].detach().cpu().numpy() if isinstance(sd[n], torch.Tensor) else sd[n], np.float32)
            a = self._map.to_lib(n, a)
            if tuple(a.shape) != tuple(s):
                raise ValueError(f"{n}: expected {self.torch_shapes()[n]}, got an array that maps to {a.shape} (library shape {s})")
            flat[o:o + a.size] = a.reshape(-1)
        with torch.no_grad():
            self.params.copy_(torch.from_numpy(flat))
        self.sync_weights()

    def sync_weights(self):
        """Re-derive the MFMA-typed weight copies after the flat parameters changed (an optimiser step on `enc.flat`).  `_forward`
        calls it by itself whenever torch's version counter of the flat buffer moved since the last call (in-place optimiser steps,
        `load_state_dict` bump it); writes that bypass the counter (`enc.flat.data...`, a raw pointer) need the explicit call."""
        _lib.check(self._lib.ishara_sync_weights(self._h, _stream()), "ishara_sync_weights")
        self._synced_version = self.flat._version

    # ------------------------------------------------------------------ forward / backward
    def _forward(self, x: torch.Tensor, training: bool, seed: Optional[int] = None) -> torch.Tensor:
        if x.dim() == 2:
            x = x[None]
        if x.shape[1:] != (self.T, self.F_in):
            raise ValueError(f"expected input [B,{self.T},{self.F_in}], got {tuple(x.shape)}")
        x = x.detach().to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        y = torch.empty((B, self.T_out, self.dim), dtype=torch.float32, device=self.device)
        if self.flat._version != self._synced_version:      # the parameters moved (optimiser step): refresh the MFMA-typed copies first
            self.sync_weights()
        self._gen += 1
        if seed is None:
            seed = (self._seed + 0x9E3779B1 * self._steps) & 0xFFFFFFFF
            self._steps += 1
        _lib.check(self._lib.ishara_encoder_forward(self._h, _lib.ptr(x), B, _lib.ptr(y), 1 if training else 0, C.c_uint32(seed), _stream()),
                   "ishara_encoder_forward")
        self._last_x = x
        return y

    def _backward(self, dy: torch.Tensor) -> torch.Tensor:
        dy = dy.detach().to(self.device, torch.float32).contiguous()
        dx = torch.empty((dy.shape[0], self.T, self.F_in), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ishara_encoder_backward(self._h, _lib.ptr(dy), dy.shape[0], _lib.ptr(dx), _stream()), "ishara_encoder_backward")
        return dx

    def _default_state(self, seed):
        return default_state_dict(self.torch_shapes(), seed)

    def _apply(self, x):
        """Training mode with grad enabled: differentiable through `_EncoderFn` — ONE live graph per encoder (the library keeps the
        saved activations and the dropout seed of its last training forward; a stale graph's backward raises).  Eval mode returns a
        constant (inference path, nothing saved): asking for a gradient w.r.t. the input there is an error, not a silent zero."""
        x = torch.as_tensor(x)
        if torch.is_grad_enabled() and self.training:
            return _EncoderFn.apply(x.to(self.device), self.flat, self)
        if torch.is_grad_enabled() and x.requires_grad:
            raise IsharaError("eval-mode forward is not differentiable (no activations are saved): call enc.train() for gradients, or "
                              "pass x.detach() / use torch.no_grad() for inference")
        return self._forward(x, training=self.training)


class ConformerEncoder(_TorchFamilyEncoder):
    """conformer/conformer.py:76-87 on MI355X.  Extra keyword arguments size the device buffers: `seq_len` (frames per clip,
    a multiple of 8, <= 512), `max_batch`, `dtype` ("bf16" storage + bf16 MFMA with fp32 accumulation, or "f32")."""

    def __init__(self, dim, num_layers=7, num_heads=8, expansion_factor=4, kernel_size=31, dropout=0.1, *,
                 seq_len=384, max_batch=64, dtype="bf16", device: Optional[str] = "cuda:0", seed=0):
        if kernel_size % 2 == 0:
            raise ValueError("kernel_size must be odd (padding=kernel_size//2 keeps the sequence length only then)")
        cfg = _lib.Config()
        cfg.family = _lib.FAMILY_TORCH_CONFORMER
        cfg.dim, cfg.num_conv_conform_blocks, cfg.num_heads = dim, num_layers, num_heads
        cfg.expansion_factor, cfg.transformer_kernel_size, cfg.dropout_rate = expansion_factor, kernel_size, dropout
        cfg.frames, cfg.features, cfg.num_classes = seq_len, dim, 60
        cfg.dtype = {"f32": _lib.F32, "bf16": _lib.BF16}[dtype]
        cfg.max_batch, cfg.max_label_len, cfg.attn_impl = max_batch, 64, 1
        self.dim, self.num_layers, self.num_heads = dim, num_layers, num_heads
        self._create(cfg, _LayoutMap(dim, num_heads), dim, seq_len, max_batch, device, seed)

    def __call__(self, x, attn_mask=None):
        """ConformerEncoder.forward(x, attn_mask=None) — conformer.py:84-87.  `attn_mask` must be None (the reference's callers
        never pass one)."""
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is not supported")
        return self._apply(x)

    forward = __call__
