"""Batch adapter for the reference's torch loader contract.

`data_loader.py`'s `ASLDataset.__getitem__` yields `(landmarks float32 [max_frames, 124, 3], List[int])`
(data_loader.py:166-195) and its DataLoader has no collate for the ragged phrases.  This module restates the
host-side steps of that file — augmentations (`:129-166`, same `random` call sequence, so a seeded
`random.Random` reproduces the reference's draws), pad / resize to `max_frames` (`:176-185`), per-sample
normalisation (`:187-188`) — and adds the collate the encoder needs: x `[B, T, F]` float32 with
F = 124*3 (flattened) or 112*2 (hands + lips x,y only: the F=224 of BASELINE config #2), y `[B, 64]` int64
padded with 59 (conv-hybrid-model.ipynb c4:21-23)."""
from __future__ import annotations

import random
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

MAX_PHRASE_LENGTH = 64   # c1:33
PAD_TOKEN_IDX = 59       # c1:5
N_FACE, N_HAND, N_POSE = 76, 21, 6          # data_loader.py:66-71 (face, left hand, right hand, pose) = 124 landmarks
LEFT_START, RIGHT_START = N_FACE, N_FACE + N_HAND


def apply_augmentations(landmarks: np.ndarray, rng: random.Random) -> np.ndarray:
    """`ASLDataset._apply_augmentations` (data_loader.py:124-166) with an explicit generator."""
    # time stretch (nearest-index resample along time)
    if rng.random() < 0.8:
        scale = rng.uniform(0.8, 1.2)
        num_frames = int(landmarks.shape[0] * scale)
        indices = np.linspace(0, landmarks.shape[0] - 1, num_frames)
        landmarks = np.stack([landmarks[int(i)] for i in indices])
    # random shift along time
    if rng.random() < 0.5:
        shift = rng.randint(-10, 10)
        if shift > 0:
            landmarks = np.pad(landmarks, ((0, shift), (0, 0), (0, 0)))[shift:, :, :]
        else:
            landmarks = np.pad(landmarks, ((-shift, 0), (0, 0), (0, 0)))[:shift, :, :]      # shift == 0 -> empty, as in the reference
    # left-right flip
    if rng.random() < 0.5:
        temp = landmarks[:, LEFT_START:LEFT_START + N_HAND].copy()
        landmarks[:, LEFT_START:LEFT_START + N_HAND] = landmarks[:, RIGHT_START:RIGHT_START + N_HAND]
        landmarks[:, RIGHT_START:RIGHT_START + N_HAND] = temp
        landmarks[:, :, 0] *= -1
    # finger dropout
    if rng.random() < 0.5:
        num_fingers = rng.randint(2, 6)
        num_windows = rng.randint(2, 3)
        for _ in range(num_windows):
            start_frame = rng.randint(0, landmarks.shape[0] - 10)
            end_frame = start_frame + rng.randint(5, 10)
            for _ in range(num_fingers):
                finger_idx = rng.randint(0, 20)
                landmarks[start_frame:end_frame, LEFT_START + finger_idx] = 0
                landmarks[start_frame:end_frame, RIGHT_START + finger_idx] = 0
    return landmarks


def pad_resize_normalize(landmarks: np.ndarray, max_frames: int = 384) -> np.ndarray:
    """data_loader.py:176-188: nearest-index resize down to max_frames or zero-pad up to it, then z-normalise
    per coordinate over (frames, landmarks)."""
    num_frames = landmarks.shape[0]
    if num_frames > max_frames:
        indices = np.linspace(0, num_frames - 1, max_frames)
        landmarks = np.stack([landmarks[int(i)] for i in indices])
    else:
        padding = np.zeros((max_frames - num_frames, landmarks.shape[1], landmarks.shape[2]))
        landmarks = np.concatenate([landmarks, padding], axis=0)
    landmarks = (landmarks - landmarks.mean(axis=(0, 1), keepdims=True)) / (landmarks.std(axis=(0, 1), keepdims=True) + 1e-8)
    return landmarks.astype(np.float32)


def to_features(landmarks: np.ndarray, layout: str = "flat") -> np.ndarray:
    """[T, 124, 3] -> [T, F].  "flat": all landmarks, xyz (F = 372).  "hands_lips_xy": the two hands + the first 70
    face landmarks, x and y only (F = 112 * 2 = 224, the feature width BASELINE config #2 is quoted on)."""
    if layout == "flat":
        return landmarks.reshape(landmarks.shape[0], -1)
    if layout == "hands_lips_xy":
        sel = np.concatenate([np.arange(LEFT_START, LEFT_START + 2 * N_HAND), np.arange(0, 70)])
        return landmarks[:, sel, :2].reshape(landmarks.shape[0], -1)
    raise ValueError(f"unknown layout {layout!r}")


def pad_phrase(phrase: Sequence[int], max_len: int = MAX_PHRASE_LENGTH, pad: int = PAD_TOKEN_IDX) -> np.ndarray:
    """tf.pad(phrase, [[0, MAX_PHRASE_LENGTH - len]], constant_values=pad_token_idx) — c4:21-22."""
    if len(phrase) > max_len:
        raise ValueError(f"phrase of {len(phrase)} tokens exceeds MAX_PHRASE_LENGTH={max_len}")
    out = np.full(max_len, pad, np.int64)
    out[:len(phrase)] = np.asarray(phrase, np.int64)
    return out


def collate(samples: Iterable[Tuple[np.ndarray, Sequence[int]]], layout: str = "flat") -> Tuple[np.ndarray, np.ndarray]:
    """List of `(landmarks [T,124,3], phrase List[int])` -> `(x [B,T,F] float32, y [B,64] int64)`: the collate the
    reference's DataLoader lacks (its default collate fails on the ragged phrases)."""
    xs, ys = [], []
    for lm, phrase in samples:
        lm = np.asarray(lm.numpy() if hasattr(lm, "numpy") else lm, np.float32)
        xs.append(to_features(lm, layout))
        ys.append(pad_phrase(phrase))
    return np.stack(xs).astype(np.float32), np.stack(ys)


class BatchAdapter:
    """Re-iterable `(x, y)` batch stream over an `ASLDataset`-shaped indexable (anything with `__len__` and
    `__getitem__ -> (landmarks, phrase)`), the `train_dataset` argument of `Model.fit`."""

    def __init__(self, dataset, batch_size: int = 64, shuffle: bool = False, seed: int = 0, layout: str = "flat",
                 drop_last: bool = False, device: Optional[str] = None):
        self.dataset, self.batch_size, self.shuffle, self.layout = dataset, batch_size, shuffle, layout
        self.drop_last, self.device = drop_last, device
        self._rng = np.random.default_rng(seed)

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple["np.ndarray", "np.ndarray"]]:
        order = np.arange(len(self.dataset))
        if self.shuffle:
            self._rng.shuffle(order)
        for i in range(0, len(order), self.batch_size):
            idx = order[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                break
            x, y = collate([self.dataset[int(j)] for j in idx], self.layout)
            if self.device is not None:
                import torch
                yield torch.from_numpy(x).to(self.device), torch.from_numpy(y).to(self.device)
            else:
                yield x, y
