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


_HAND_SWAP = np.arange(N_FACE + 2 * N_HAND + N_POSE)
_HAND_SWAP[LEFT_START:LEFT_START + N_HAND] = np.arange(RIGHT_START, RIGHT_START + N_HAND)
_HAND_SWAP[RIGHT_START:RIGHT_START + N_HAND] = np.arange(LEFT_START, LEFT_START + N_HAND)


def _resample_time(clip: np.ndarray, length: int) -> np.ndarray:
    """Nearest-below resample along time: frame j of the result is frame floor(j * (T - 1) / (length - 1))."""
    return clip[np.linspace(0, clip.shape[0] - 1, length).astype(np.int64)]


def apply_augmentations(landmarks: np.ndarray, rng: random.Random) -> np.ndarray:
    """The four augmentations of `ASLDataset._apply_augmentations` (data_loader.py:124-166) on one clip `[T,124,3]`,
    drawing from `rng` in the reference's order (gate, then that augmentation's parameters), so a seeded
    `random.Random` reproduces its output.  Returns a new array; the input is not modified."""
    clip = np.array(landmarks)
    if rng.random() < 0.8:                                        # time stretch by 0.8x-1.2x
        clip = _resample_time(clip, int(clip.shape[0] * rng.uniform(0.8, 1.2)))
    if rng.random() < 0.5:                                        # shift by up to 10 frames, zero fill
        s, n = rng.randint(-10, 10), clip.shape[0]
        moved = np.zeros_like(clip)
        if s > 0:
            moved[:max(n - s, 0)] = clip[s:]
        elif s < 0:
            moved[-s:] = clip[:max(n + s, 0)]
        else:
            moved = moved[:0]                                     # the reference's `[:shift]` with shift == 0 is empty
        clip = moved
    if rng.random() < 0.5:                                        # mirror: swap the hands, negate x
        clip = clip[:, _HAND_SWAP]
        clip[..., 0] = -clip[..., 0]
    if rng.random() < 0.5:                                        # finger dropout in 2-3 short windows
        n_fingers, n_windows = rng.randint(2, 6), rng.randint(2, 3)
        for _ in range(n_windows):
            t0 = rng.randint(0, clip.shape[0] - 10)
            t1 = t0 + rng.randint(5, 10)
            fingers = np.array([rng.randint(0, 20) for _ in range(n_fingers)])
            clip[t0:t1, np.concatenate([LEFT_START + fingers, RIGHT_START + fingers])] = 0
    return clip


def pad_resize_normalize(landmarks: np.ndarray, max_frames: int = 384) -> np.ndarray:
    """data_loader.py:176-188: clips longer than `max_frames` are resampled down (nearest-below), shorter ones
    zero-padded at the end; then each coordinate is z-normalised over (frames, landmarks), eps 1e-8 on the std."""
    n = landmarks.shape[0]
    if n > max_frames:
        clip = _resample_time(landmarks, max_frames).astype(np.float64)
    else:
        clip = np.zeros((max_frames,) + landmarks.shape[1:], np.float64)
        clip[:n] = landmarks
    mu, sd = clip.mean(axis=(0, 1), keepdims=True), clip.std(axis=(0, 1), keepdims=True)
    return ((clip - mu) / (sd + 1e-8)).astype(np.float32)


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
