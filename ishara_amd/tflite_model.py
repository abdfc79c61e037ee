"""TFLite-shaped inference wrapper — `TFLiteModel` of conv-hybrid-model.ipynb c13:1-25 and the `serving_default`
signature the reference exports (c14, c16:10-14): input `inputs` float32 [n_frames, 276] raw landmarks with NaNs,
output {'outputs': float32 one-hot [n_chars, 59]}.

Device side (all in libishara_hip.so): preprocessing kernel (frame filter, resize/pad, normalise, NaN->0) -> encoder
forward at B=1 -> greedy decode, captured ONCE into a hipGraph (torch.cuda.CUDAGraph only records the launches the
library makes on the capture stream) and replayed per clip.  Host side: the len<3 fallback phrase and one_hot(., 59)
(c13:22-24), as in the reference wrapper.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .model import Model, _stream

N_COLS = 276
# concat order of pre_process1 (c3:111) with landmark counts
PARTS = [("lip", 40), ("rhand", 21), ("lhand", 21), ("rpose", 5), ("lpose", 5)]
FALLBACK_PHRASE = np.array([17, 0, 32, 12, 36, 0, 12, 32, 49, 46, 36], dtype=np.int64)     # c13:22-23


class TFLiteModel:
    def __init__(self, model: Model, stats: Optional[Dict[str, tuple]] = None, max_frames: int = 1024, use_graph: bool = True):
        if model.F != N_COLS:
            raise ValueError(f"the TFLite wrapper feeds {N_COLS} columns (92 landmarks x 3); model has F={model.F}")
        self.model, self.max_frames, self.T = model, max_frames, model.T
        dev = model.device
        mean = np.concatenate([(stats[n][0] if stats else np.zeros((c, 3), np.float32)).reshape(-1) for n, c in PARTS])
        std = np.concatenate([(stats[n][1] if stats else np.ones((c, 3), np.float32)).reshape(-1) for n, c in PARTS])
        self._mean = torch.from_numpy(mean.astype(np.float32)).to(dev)
        self._std = torch.from_numpy(std.astype(np.float32)).to(dev)
        self._raw = torch.zeros((max_frames, N_COLS), dtype=torch.float32, device=dev)
        self._n = torch.zeros(1, dtype=torch.int32, device=dev)
        self._x = torch.zeros((1, self.T, N_COLS), dtype=torch.float32, device=dev)
        self._logits = torch.zeros((1, self.T, model.C), dtype=torch.float32, device=dev)
        self._idx = torch.zeros((1, self.T), dtype=torch.int32, device=dev)
        self._len = torch.zeros(1, dtype=torch.int32, device=dev)
        self._graph = None
        if use_graph:
            self._capture()

    def _launch(self):
        lib, m = self.model._lib, self.model
        _lib.check(lib.ishara_preprocess(_lib.ptr(self._raw), _lib.ptr(self._n), self.max_frames, _lib.ptr(self._mean), _lib.ptr(self._std),
                                         _lib.ptr(self._x), self.T, _stream()), "ishara_preprocess")
        _lib.check(lib.ishara_forward(m._h, _lib.ptr(self._x), 1, _lib.ptr(self._logits), 0, C.c_uint32(0), _stream()), "ishara_forward")
        _lib.check(lib.ishara_greedy_decode(_lib.ptr(self._logits), 1, self.T, m.C, m.C - 1, _lib.ptr(self._idx), _lib.ptr(self._len), _stream()),
                   "ishara_greedy_decode")

    def _capture(self):
        side = torch.cuda.Stream(device=self.model.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._launch()                       # warm-up outside capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._launch()
        self._graph = g

    def predict_indices(self, inputs) -> np.ndarray:
        x = np.asarray(inputs, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != N_COLS:
            raise ValueError(f"inputs must be [n_frames, {N_COLS}]")
        n = x.shape[0]
        if n > self.max_frames:
            raise ValueError(f"clip of {n} frames exceeds max_frames={self.max_frames}")
        if n:
            self._raw[:n].copy_(torch.from_numpy(np.ascontiguousarray(x)), non_blocking=False)
        self._n.fill_(n)
        if self._graph is not None:
            self._graph.replay()
        else:
            self._launch()
        ln = int(self._len.item())
        return self._idx[0, :ln].cpu().numpy().astype(np.int64)

    def __call__(self, inputs) -> Dict[str, np.ndarray]:
        idx = self.predict_indices(inputs)
        if idx.shape[0] < 3:                                  # c13:22-23
            idx = FALLBACK_PHRASE
        out = np.zeros((idx.shape[0], 59), dtype=np.float32)  # tf.one_hot(x, 59): index 59 -> zero row
        ok = idx < 59
        out[np.arange(idx.shape[0])[ok], idx[ok]] = 1.0
        return {"outputs": out}

    # reference spelling: interpreter.get_signature_runner("serving_default")(inputs=frame)  (c16:10-13)
    def get_signature_runner(self, name: str = "serving_default"):
        if name != "serving_default":
            raise KeyError(name)
        return lambda inputs: self(inputs)
