#!/usr/bin/env python3
"""Golden vectors for ishara_amd/data.py from the reference's own data_loader.py (imported in THIS container only;
it never travels).  `ASLDataset` is instantiated without its file-reading __init__ and fed synthetic landmarks;
`random.seed(k)` fixes the augmentation draws.  Writes tests/golden/dataloader_adapter.npz."""
import os, random, sys
import numpy as np
import pandas as pd

REF = "/root/reference"
sys.path.insert(0, REF)
import data_loader as ref  # noqa: E402

out = {}
g = np.random.default_rng(0)
cases = [(30, 64), (100, 64), (64, 64), (20, 32)]     # (frames, max_frames): pad, resize, exact fit, short
for ci, (frames, max_frames) in enumerate(cases):
    lm = g.standard_normal((frames, 124, 3)).astype(np.float32).astype(np.float64)   # float32-representable: the exact ops below stay so
    for seed in ((1, 2, 3, 4) if ci in (0, 3) else ()):       # augmentation draws: the two short clips keep the fixture small
        ds = object.__new__(ref.ASLDataset)
        ds.augment = True
        random.seed(100 * ci + seed)
        aug = ds._apply_augmentations(lm.copy())
        out[f"aug_out_{ci}_{seed}"] = aug.astype(np.float32)
    out[f"aug_in_{ci}"] = lm.astype(np.float32)
    # __getitem__ without augmentation: pad / resize + normalise + phrase encoding
    ds = object.__new__(ref.ASLDataset)
    ds.augment = False
    ds.max_frames = max_frames
    ds.metadata = pd.DataFrame({"phrase": ["ab c"], "sequence_id": [0]})
    ds.char_to_pred = {"a": 3, "b": 7, " ": 0, "c": 11}
    ds._load_landmarks = lambda idx, lm=lm: lm.copy()
    x, phrase = ds[0]
    out[f"item_x_{ci}"] = x.numpy()
    out[f"item_phrase_{ci}"] = np.asarray(phrase, np.int64)
    out[f"item_maxframes_{ci}"] = np.asarray(max_frames)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "dataloader_adapter.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, len(out), "arrays")
