"""Host-side mirror of the reference's Keras object surface for the hot path.

`get_model(...)` has the signature of `conv-hybrid-model.ipynb c7:1-11` (plus the notebook
globals INPUT_SHAPE / len(char_to_num) as explicit kwargs) and returns a `Model` exposing what
the reference notebooks touch: `model(x, training=...)`, `compile`, `fit` with Keras-style
callbacks, `optimizer.learning_rate / .weight_decay`, `save_weights`, `summary`.

All arithmetic runs in libishara_hip.so (hand-written HIP for gfx950) through the C ABI in
include/ishara_hip.h; torch-ROCm tensors are only containers for device memory and the source
of the current stream.  There is no CPU path: constructing a Model with a device requires the
built library and a GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple,  Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import IsharaError

PAD_TOKEN_IDX = 59  # c1:5


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Scalar:
    """Mimics the `tf.Variable` hypers the reference callbacks read (`.numpy()`, c11:64-65)."""

    def __init__(self, v): self.v = float(v)
    def numpy(self): return np.float32(self.v)
    def assign(self, v): self.v = float(v.v if isinstance(v, _Scalar) else v)
    def __float__(self): return self.v
    def __mul__(self, o): return _Scalar(self.v * float(o))
    __rmul__ = __mul__
    def __repr__(self): return f"{self.v:g}"


class Optimizer:
    """Lookahead(RectifiedAdam(sma_threshold=4), sync_period=5) — c7:68-69.  The update itself is
    one fused HIP kernel over the flat parameter buffer (csrc/optimizer.hip)."""

    def __init__(self, learning_rate=1e-3, weight_decay=0.0):
        self._lr = _Scalar(learning_rate)
        # The reference assigns `.weight_decay` on the Lookahead wrapper (c11:64), which is not
        # one of its hyper-parameters, so RectifiedAdam keeps weight_decay=0.
        # `apply_weight_decay=True` turns the assigned value into a real decoupled decay.
        self._wd = _Scalar(weight_decay)
        self.apply_weight_decay = False
        self.iterations = 0

    @property
    def learning_rate(self): return self._lr
    @learning_rate.setter
    def learning_rate(self, v): self._lr.assign(v)
    lr = learning_rate
    @property
    def weight_decay(self): return self._wd
    @weight_decay.setter
    def weight_decay(self, v): self._wd.assign(v)


class History:
    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []


class Callback:
    """Keras-style callback base (on_epoch_begin/end receive (epoch, logs))."""
    model = None
    def set_model(self, model): self.model = model
    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_train_batch_end(self, batch, logs=None): pass


class LearningRateScheduler(Callback):
    """tf.keras.callbacks.LearningRateScheduler(fn(epoch) -> lr) — c11:55."""

    def __init__(self, schedule, verbose=0):
        self.schedule, self.verbose = schedule, verbose

    def on_epoch_begin(self, epoch, logs=None):
        lr = float(self.schedule(epoch))
        self.model.optimizer.learning_rate = lr
        if self.verbose:
            print(f"Epoch {epoch + 1}: LearningRateScheduler setting learning rate to {lr}.")


def lrfn(current_step, num_warmup_steps, lr_max, num_cycles=0.50, num_training_steps=50, warmup_method="exp"):
    """c11:1-11."""
    if current_step < num_warmup_steps:
        if warmup_method == "log":
            return lr_max * 0.10 ** (num_warmup_steps - current_step)
        return lr_max * 2 ** -(num_warmup_steps - current_step)
    progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))) * lr_max


def _keras_init(name: str, shape, g: np.random.Generator) -> np.ndarray:
    """Keras default initialisers for the layers of c5/c7 (glorot_uniform kernels, zeros biases,
    (1,0) norms, BatchNorm moving (0,1))."""
    leaf = name.rsplit("/", 1)[-1]
    if leaf in ("bias", "beta", "moving_mean"):
        return np.zeros(shape, np.float32)
    if leaf in ("gamma", "moving_variance"):
        return np.ones(shape, np.float32)
    if leaf == "depthwise_kernel" or name.endswith("depthwise_conv/kernel"):
        k, c = shape
        lim = math.sqrt(6.0 / (k * c + k))
    elif name.endswith("_eca/kernel"):
        lim = math.sqrt(6.0 / 10.0)
    else:
        lim = math.sqrt(6.0 / (shape[0] + shape[1]))
    return g.uniform(-lim, lim, size=shape).astype(np.float32)


class Model:
    def __init__(self, cfg: _lib.Config, device: Optional[str] = "cuda:0", seed: int = 0):
        self._lib = _lib.load()
        self._cfg = cfg
        self._h = C.c_void_p()
        _lib.check(self._lib.ishara_create(C.byref(cfg), C.byref(self._h)), "ishara_create")
        self.T, self.F, self.C = cfg.frames, cfg.features, cfg.num_classes
        self.max_batch = cfg.max_batch
        self.n_total = int(self._lib.ishara_param_total(self._h))
        self.n_train = int(self._lib.ishara_param_trainable(self._h))
        self.entries = []
        for i in range(self._lib.ishara_param_entries(self._h)):
            name, nd, sh, off, tr = C.c_char_p(), C.c_int32(), (C.c_int64 * 2)(), C.c_int64(), C.c_int32()
            _lib.check(self._lib.ishara_param_info(self._h, i, C.byref(name), C.byref(nd), C.byref(sh), C.byref(off), C.byref(tr)))
            shape = (int(sh[0]),) if nd.value == 1 else (int(sh[0]), int(sh[1]))
            self.entries.append((name.value.decode(), shape, int(off.value), bool(tr.value)))
        self.optimizer = Optimizer()
        self.loss = "ctc"
        self.stop_training = False
        self.device = None
        self._step_seed = seed * 7919 + 17
        self._steps = 0
        if device is not None:
            self._to_device(device, seed)

    # ------------------------------------------------------------------ device state
    def _to_device(self, device, seed):
        if not torch.cuda.is_available():
            raise IsharaError("ishara_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        dev = self.device
        self.params = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.opt_m = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.opt_v = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.opt_slow = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        wsb = int(self._lib.ishara_workspace_bytes(self._h))
        self.workspace = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = self.workspace.data_ptr() + off
        _lib.check(self._lib.ishara_bind(self._h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.opt_m),
                                         _lib.ptr(self.opt_v), _lib.ptr(self.opt_slow), C.c_void_p(self._ws_ptr), wsb), "ishara_bind")
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self._nll_buf = torch.zeros(self.max_batch, dtype=torch.float32, device=dev)
        g = np.random.default_rng(seed)
        self.set_weights({name: _keras_init(name, shape, g) for name, shape, _, _ in self.entries})

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ishara_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def get_weights(self) -> Dict[str, np.ndarray]:
        flat = self.params.detach().cpu().numpy()
        return {n: flat[o:o + int(np.prod(s))].reshape(s).copy() for n, s, o, _ in self.entries}

    def get_gradients(self) -> Dict[str, np.ndarray]:
        flat = self.grads.detach().cpu().numpy()
        return {n: flat[o:o + int(np.prod(s))].reshape(s).copy() for n, s, o, t in self.entries if t}

    def set_weights(self, weights: Dict[str, np.ndarray], reset_optimizer: bool = True):
        flat = self.params.detach().cpu().numpy().copy()
        for n, s, o, _ in self.entries:
            if n in weights:
                w = np.asarray(weights[n], dtype=np.float32)
                if tuple(w.shape) != tuple(s):
                    raise ValueError(f"{n}: expected shape {s}, got {w.shape}")
                flat[o:o + w.size] = w.reshape(-1)
        self.params.copy_(torch.from_numpy(flat))
        if reset_optimizer:
            self.opt_m.zero_(); self.opt_v.zero_()
            self.opt_slow.copy_(self.params[:self.n_train])     # Lookahead slow weights start at theta_0
            self.optimizer.iterations = 0
            self._lib.ishara_optimizer_set_iterations(self._h, 0)
        _lib.check(self._lib.ishara_sync_weights(self._h, _stream()), "ishara_sync_weights")

    def save_weights(self, path: str):
        """model.save_weights (c9:10).  `*.h5` / `*.hdf5`: the Keras-2 save_weights HDF5 layout, written through the image's HDF5
        library (keras_h5.py); anything else: .npz keyed by the library's Keras-style names."""
        if path.endswith((".h5", ".hdf5")):
            from . import keras_h5
            keras_h5.save_weights_h5(path, self.get_weights(), [(n, tuple(s)) for n, s, _, _ in self.entries])
            return
        np.savez(path if path.endswith(".npz") else path + ".npz", **self.get_weights())

    def load_weights(self, path: str):
        """model.load_weights: a Keras-2 `.h5` weights file (matched by layer / weight order and shape, like by_name=False) or the .npz."""
        if path.endswith((".h5", ".hdf5")):
            from . import keras_h5
            self.set_weights(keras_h5.load_weights_h5(path, [(n, tuple(s)) for n, s, _, _ in self.entries]))
            return
        with np.load(path if path.endswith(".npz") else path + ".npz") as z:
            self.set_weights({k: z[k] for k in z.files})

    def save_keras_weights(self, path: str):
        """Ordered list of `keras_model.get_weights()` as arr_0.. (keras_interchange.py)."""
        from . import keras_interchange as K
        K.save_keras_npz(path, self.get_weights(), [(n, tuple(s)) for n, s, _, _ in self.entries])

    def load_keras_weights(self, path: str):
        from . import keras_interchange as K
        self.set_weights(K.load_keras_npz(path, [(n, tuple(s)) for n, s, _, _ in self.entries]))

    def count_params(self): return self.n_total

    def summary(self, print_fn=print):
        """model.summary() (c7:83), grouped per layer prefix."""
        groups: Dict[str, int] = {}
        for n, s, _, _ in self.entries:
            top = n.split("/")[0]
            groups[top] = groups.get(top, 0) + int(np.prod(s))
        print_fn('Model: "ishara_hip"')
        print_fn("=" * 65)
        for k, v in groups.items():
            print_fn(f" {k:<44s}{v:>12,d}")
        print_fn("=" * 65)
        print_fn(f"Total params: {self.n_total:,}")
        print_fn(f"Trainable params: {self.n_train:,}")
        print_fn(f"Non-trainable params: {self.n_total - self.n_train:,}")

    # ------------------------------------------------------------------ forward
    def _as_input(self, x) -> torch.Tensor:
        x = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
        if x.dim() == 2:
            x = x[None]
        if x.shape[1:] != (self.T, self.F):
            raise ValueError(f"expected input [B,{self.T},{self.F}], got {tuple(x.shape)}")
        return x.to(self.device, torch.float32).contiguous()

    def __call__(self, x, training: bool = False, seed: Optional[int] = None) -> torch.Tensor:
        x = self._as_input(x)
        B = x.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        logits = torch.empty((B, self.T, self.C), dtype=torch.float32, device=self.device)
        if seed is None:
            seed = (self._step_seed + 0x9E3779B1 * self._steps) & 0xFFFFFFFF
        _lib.check(self._lib.ishara_forward(self._h, _lib.ptr(x), B, _lib.ptr(logits), 1 if training else 0,
                                            C.c_uint32(seed), _stream()), "ishara_forward")
        self._last_x = x          # keep the input alive: the stem weight gradient re-reads it
        return logits

    predict = __call__

    def compile(self, loss=None, optimizer=None):
        """model.compile(loss=CTCLoss, optimizer=Lookahead(RAdam)) — c7:70.  Only that pair exists
        in the HIP library; `loss`/`optimizer` are accepted for signature parity."""
        if optimizer is not None and isinstance(optimizer, Optimizer):
            self.optimizer = optimizer
        return self

    # ------------------------------------------------------------------ training
    def loss_and_gradients(self, x, y, seed: Optional[int] = None, loss_scale: float = 1.0):
        """forward(training=True) + CTCLoss + backward.  Returns (loss tensor[1], logits)."""
        logits = self(x, training=True, seed=seed)
        y = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y).to(self.device, torch.int64).contiguous()
        B = logits.shape[0]
        if y.shape != (B, self._cfg.max_label_len):
            raise ValueError(f"labels must be [B,{self._cfg.max_label_len}] padded with {self.C - 1}")
        _lib.check(self._lib.ishara_loss_backward(self._h, _lib.ptr(logits), _lib.ptr(y), B, _lib.ptr(self._loss_buf),
                                                  _lib.ptr(self._nll_buf), C.c_float(loss_scale), _stream()), "ishara_loss_backward")
        return self._loss_buf, logits

    def apply_gradients(self):
        wd = float(self.optimizer.weight_decay) if self.optimizer.apply_weight_decay else 0.0
        _lib.check(self._lib.ishara_optimizer_step(self._h, C.c_float(float(self.optimizer.learning_rate)), C.c_float(wd), _stream()),
                   "ishara_optimizer_step")
        self.optimizer.iterations += 1
        self._steps += 1

    def train_on_batch(self, x, y, seed: Optional[int] = None, allreduce: bool = True) -> torch.Tensor:
        """One Keras train_step (c12): forward, CTC, backward, [RCCL grad all-reduce], update.
        Returns the device loss tensor (no host sync).  `allreduce=False` skips the gradient exchange (bench.py's diagnostic
        of the exposed all-reduce time; replicas diverge — never for training)."""
        from . import parallel
        world = parallel.world_size()
        if not allreduce:
            loss, _ = self.loss_and_gradients(x, y, seed=seed, loss_scale=1.0 / world)
            self.apply_gradients()
            return loss
        overlap = world > 1 and parallel.overlap_enabled()
        if overlap:
            self.enable_grad_buckets()
        if seed is None and world > 1:
            # replicas share the weight-initialisation seed but draw independent dropout / drop-path masks for their shards,
            # like the reference's replicas (tf.distribute / nn.DataParallel keep per-replica RNG streams)
            seed = ((self._step_seed + 0x9E3779B1 * self._steps) ^ (parallel.rank() * 0x85EBCA6B)) & 0xFFFFFFFF
        loss, _ = self.loss_and_gradients(x, y, seed=seed, loss_scale=1.0 / world)
        if overlap:
            parallel.allreduce_buckets_(self)          # ranges of the gradient reduced on a side stream as the backward pass finishes them
        elif world > 1:
            parallel.allreduce_sum_(self.grads[:self.n_train])
        self.apply_gradients()
        return loss

    # ---- gradient buckets (all-reduce overlapped with the backward pass; opt-in, see parallel.overlap_enabled)
    def grad_buckets(self) -> List[Tuple[int, int]]:
        """(offset, count) ranges of the flat gradient in the order the backward pass completes them."""
        out = []
        for i in range(int(self._lib.ishara_grad_buckets(self._h))):
            off, cnt = C.c_int64(), C.c_int64()
            _lib.check(self._lib.ishara_grad_bucket(self._h, i, C.byref(off), C.byref(cnt)), "ishara_grad_bucket")
            out.append((int(off.value), int(cnt.value)))
        return out

    def enable_grad_buckets(self):
        _lib.check(self._lib.ishara_grad_buckets_enable(self._h), "ishara_grad_buckets_enable")

    def wait_grad_bucket(self, i: int, side_stream: "torch.cuda.Stream"):
        """Makes `side_stream` wait until bucket i of the last backward pass is final (no host sync)."""
        _lib.check(self._lib.ishara_grad_bucket_wait(self._h, i, C.c_void_p(side_stream.cuda_stream)), "ishara_grad_bucket_wait")

    def fit(self, train_dataset: Iterable, validation_data: Optional[Iterable] = None, epochs: int = 1,
            callbacks: Sequence[Callback] = (), steps_per_epoch: Optional[int] = None, verbose: int = 1) -> History:
        """model.fit(train_dataset, validation_data=, epochs=, callbacks=[...]) — c12:1-10.
        Datasets are re-iterable objects yielding (x [B,T,F] float32, y [B,64] int64)."""
        hist = History()
        for cb in callbacks:
            if hasattr(cb, "set_model"): cb.set_model(self)
            elif not getattr(cb, "model", None): cb.model = self
        for cb in callbacks: getattr(cb, "on_train_begin", lambda logs=None: None)()
        self.stop_training = False
        for epoch in range(epochs):
            logs: Dict[str, float] = {}
            for cb in callbacks: cb.on_epoch_begin(epoch, logs)
            tot = torch.zeros(1, dtype=torch.float32, device=self.device)
            n = 0
            for bi, (x, y) in enumerate(train_dataset):
                if steps_per_epoch is not None and bi >= steps_per_epoch:
                    break
                tot += self.train_on_batch(x, y)
                n += 1
                for cb in callbacks: getattr(cb, "on_train_batch_end", lambda b, logs=None: None)(bi, logs)
            logs["loss"] = float(tot.item()) / max(n, 1)
            logs["lr"] = float(self.optimizer.learning_rate)
            if validation_data is not None:
                logs["val_loss"] = self.evaluate(validation_data)
            for cb in callbacks: cb.on_epoch_end(epoch, logs)
            hist.epoch.append(epoch)
            for k, v in logs.items(): hist.history.setdefault(k, []).append(v)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4g}" for k, v in logs.items()))
            if self.stop_training:
                break
        for cb in callbacks: getattr(cb, "on_train_end", lambda logs=None: None)()
        return hist

    def profile_step(self, x, y, seed: Optional[int] = None) -> Dict[str, dict]:
        """One training step with every kernel launch bracketed by HIP events on the launch stream.
        Returns {kernel family: {launches, ms, bytes (algorithmic), flops}}."""
        _lib.check(self._lib.ishara_profile_enable(self._h, 1))
        try:
            self.train_on_batch(x, y, seed=seed)
            buf = C.create_string_buffer(1 << 16)
            n = self._lib.ishara_profile_report(self._h, buf, len(buf))
            if n < 0:
                _lib.check(n, "ishara_profile_report")
        finally:
            self._lib.ishara_profile_enable(self._h, 0)
        out = {}
        for line in buf.value.decode().splitlines():
            k, cnt, ms, by, fl = line.split()
            out[k] = dict(launches=int(cnt), ms=float(ms), bytes=float(by), flops=float(fl))
        return out

    def ctc_loss(self, y, logits) -> torch.Tensor:
        """CTCLoss(labels, logits) (c6:1-13) on the GPU; returns per-sample nll [B]."""
        logits = logits.to(self.device, torch.float32).contiguous()
        y = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y).to(self.device, torch.int64).contiguous()
        B, T, Cc = logits.shape
        L = y.shape[1]
        ws = torch.empty(int(self._lib.ishara_ctc_workspace_bytes(B, T, L)), dtype=torch.uint8, device=self.device)
        nll = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ishara_ctc_loss(_lib.ptr(logits), _lib.ptr(y), B, T, Cc, L, Cc - 1, _lib.ptr(nll), None,
                                             C.c_float(1.0), _lib.ptr(ws), _stream()), "ishara_ctc_loss")
        return nll

    def evaluate(self, dataset: Iterable) -> float:
        tot, n = 0.0, 0
        for x, y in dataset:
            logits = self(x, training=False)
            tot += float(self.ctc_loss(y, logits).mean().item())
            n += 1
        return tot / max(n, 1)

    # ------------------------------------------------------------------ decode
    def decode_batch(self, logits: torch.Tensor) -> List[np.ndarray]:
        """decode_batch_predictions (c8:15-20) -> list of index arrays (decode_phrase, c8:4-12)."""
        logits = logits.to(self.device, torch.float32).contiguous()
        B, T, Cc = logits.shape
        idx = torch.empty((B, T), dtype=torch.int32, device=self.device)
        ln = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.ishara_greedy_decode(_lib.ptr(logits), B, T, Cc, Cc - 1, _lib.ptr(idx), _lib.ptr(ln), _stream()),
                   "ishara_greedy_decode")
        idx, ln = idx.cpu().numpy(), ln.cpu().numpy()
        return [idx[b, :ln[b]].astype(np.int64) for b in range(B)]


def make_config(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=(11, 5, 3),
                num_conv_per_block=3, dropout_rate=0.2, num_heads=8, expansion_factor=2, transformer_kernel_size=15,
                input_shape=(176, 276), num_classes=60, top_dim=0, squeeze_expansion=0, conformer_expansion=0,
                head_dropout=0.4, conformer_attn_dropout=0.1, dtype="bf16", max_batch=64, max_label_len=64,
                attn_impl=1) -> _lib.Config:
    c = _lib.Config()
    c.dim, c.num_conv_squeeze_blocks, c.num_conv_conform_blocks = dim, num_conv_squeeze_blocks, num_conv_conform_blocks
    ks = list(kernel_sizes)
    c.num_kernel_sizes = len(ks)
    for i, k in enumerate(ks[:8]):
        c.kernel_sizes[i] = k
    c.num_conv_per_block, c.dropout_rate, c.num_heads = num_conv_per_block, dropout_rate, num_heads
    c.expansion_factor, c.transformer_kernel_size = expansion_factor, transformer_kernel_size
    c.frames, c.features, c.num_classes = input_shape[0], input_shape[1], num_classes
    c.top_dim, c.squeeze_expansion, c.conformer_expansion = top_dim, squeeze_expansion, conformer_expansion
    c.head_dropout, c.conformer_attn_dropout = head_dropout, conformer_attn_dropout
    c.dtype = {"f32": _lib.F32, "fp32": _lib.F32, "float32": _lib.F32, "bf16": _lib.BF16, "bfloat16": _lib.BF16,
               "f16": _lib.F16, "fp16": _lib.F16, "float16": _lib.F16}[dtype]
    c.max_batch, c.max_label_len, c.attn_impl = max_batch, max_label_len, attn_impl
    return c


def get_model(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3],
              num_conv_per_block=3, dropout_rate=0.2, num_heads=8, expansion_factor=2, transformer_kernel_size=15,
              *, input_shape=(176, 276), num_classes=60, dtype="bf16", max_batch=64, device="cuda:0", seed=0, **variant):
    """get_model(...) of conv-hybrid-model.ipynb c7:1-72.  The positional/keyword arguments are
    the reference's; the keyword-only ones replace the notebook globals (INPUT_SHAPE c3:119,
    len(char_to_num) c1:7) and pick the build's storage dtype / workspace size."""
    cfg = make_config(dim, num_conv_squeeze_blocks, num_conv_conform_blocks, kernel_sizes, num_conv_per_block, dropout_rate,
                      num_heads, expansion_factor, transformer_kernel_size, input_shape, num_classes, dtype=dtype,
                      max_batch=max_batch, **variant)
    model = Model(cfg, device=device, seed=seed)
    model.compile(loss="ctc", optimizer=Optimizer())
    return model
