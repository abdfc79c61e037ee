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

# MediaPipe landmark ids behind the 92 selected landmarks (dataset metadata, c1:12-22): 40 lip points of the face mesh, the arm /
# hand points of the pose model for the left (13..21 odd) and right (14..22 even) side
LIP_IDS = (61, 185, 40, 39, 37, 0, 267, 269, 270, 409, 291, 146, 91, 181, 84, 17, 314, 405, 321, 375,
           78, 191, 80, 81, 82, 13, 312, 311, 310, 415, 95, 88, 178, 87, 14, 317, 402, 318, 324, 308)
LPOSE_IDS, RPOSE_IDS = tuple(range(13, 22, 2)), tuple(range(14, 23, 2))


def selected_columns():
    """SEL_COLS of the reference (c1:24-28): the parquet column names the 276 input features are read from — one block per axis
    (x | y | z), inside a block right hand 0..20, left hand 0..20, left pose, right pose, lips."""
    per_axis = ([("right_hand", i) for i in range(21)] + [("left_hand", i) for i in range(21)]
                + [("pose", i) for i in LPOSE_IDS + RPOSE_IDS] + [("face", i) for i in LIP_IDS])
    return [f"{axis}_{part}_{i}" for axis in "xyz" for part, i in per_axis]


def write_inference_args(path: str = "inference_args.json") -> str:
    """The side file the reference ships next to `model.tflite` (c14:9-10) and reads back in c15:1-2:
    `{"selected_columns": SEL_COLS}`."""
    import json
    with open(path, "w") as f:
        json.dump({"selected_columns": selected_columns()}, f)
    return path


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

    def export(self, directory: str = ".", weights: str = "model.h5") -> Dict[str, str]:
        """The export step of c14:1-10 as far as this build goes: the model's weights (`model.h5`, Keras-2 `save_weights` layout; use a
        `.npz` name where libhdf5 is missing) and `inference_args.json`.  The fp16 numerics of the reference's `.tflite` are what
        `get_model(dtype="f16")` runs; the flatbuffer container itself is not written (nothing in this image can validate one)."""
        import os
        os.makedirs(directory, exist_ok=True)
        wpath = os.path.join(directory, weights)
        self.model.save_weights(wpath)
        return {"weights": wpath, "inference_args": write_inference_args(os.path.join(directory, "inference_args.json"))}

    # reference spelling: interpreter.get_signature_runner("serving_default")(inputs=frame)  (c16:10-13)
    def get_signature_runner(self, name: str = "serving_default"):
        if name != "serving_default":
            raise KeyError(name)
        return lambda inputs: self(inputs)
