"""Weight interchange with the reference's Keras model.

`model.save_weights("model.h5")` (conv-hybrid-model.ipynb c9:10): the HDF5 file itself is handled by `keras_h5.py` (through the image's
HDF5 C library); what goes INTO it — and what moves without HDF5 — is the ORDERED LIST that Keras itself defines:
`model.get_weights()` / `model.set_weights(list)`.  On the reference side (a maintainer with TensorFlow):

    np.savez("ishara_keras_weights.npz", *model.get_weights())          # export -> arr_0, arr_1, ...
    model.set_weights([z[f"arr_{i}"] for i in range(len(z.files))])     # import

This module gives the position and Keras shape of every parameter of `ishara_amd.Model` in that list.

Order (Keras-2 `Model.weights`): the functional model's layers in creation order, each contributing
`layer.trainable_weights + layer.non_trainable_weights`; a subclassed layer (SqueezeformerBlock, ConformerBlock and
their sub-modules, c5) lists its tracked sub-layers in attribute-assignment order.  The library's own parameter order
(`Model.entries`) already follows that rule layer by layer; the one difference is the ConformerBlock, a SINGLE Keras
layer whose BatchNormalization moving statistics therefore come after ALL of the block's trainable weights
(`conformer_i/conv/batch_norm/moving_*` move behind `conformer_i/layer_norm2/beta`).
Shapes: DepthwiseConv1D kernels are `[k, C, 1]`, the grouped `Conv1D(groups=C)` kernel `[k, 1, C]`, 1x1 `Conv1D`
kernels `[1, in, out]`, the ECA `Conv1D(1, 5)` kernel `[5, 1, 1]`; everything else as stored.

Parity of this ordering is UNPINNED (no TensorFlow here, no saved .h5 in the reference); what is checked
(tests/test_keras_interchange.py) is the round trip and the per-layer parameter counts of the reference's saved
`model.summary()` outputs (tests/golden/structural_pins.json)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

Entry = Tuple[str, Tuple[int, ...]]      # (library parameter name, library shape)


def _keras_shape(name: str, shape: Tuple[int, ...]) -> Tuple[int, ...]:
    if name.endswith("/depthwise_kernel"):                                   # DepthwiseConv1D: [k, C, 1]
        return (shape[0], shape[1], 1)
    if "/depthwise_conv/kernel" in name:                                     # Conv1D(groups=C): [k, 1, C]
        return (shape[0], 1, shape[1])
    if name.endswith("_eca/kernel"):                                         # Conv1D(1, 5) on the channel axis: [5, 1, 1]
        return (shape[0], 1, 1)
    if name.endswith("/kernel") and any(s in name for s in ("/conv1/", "/conv3/", "/pointwise_conv1/", "/pointwise_conv2/")):
        return (1, shape[0], shape[1])                                       # Conv1D(kernel_size=1): [1, in, out]
    return tuple(shape)


def keras_weight_order(entries: Sequence[Entry]) -> List[Tuple[str, Tuple[int, ...]]]:
    """[(library name, Keras shape)] in `model.get_weights()` order.  `entries`: (name, shape) in library order
    (`[(n, s) for n, s, _, _ in model.entries]`)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    held: List[Tuple[str, Tuple[int, ...]]] = []
    block = None
    for name, shape in entries:
        top = name.split("/")[0]
        if block is not None and top != block:          # leaving a ConformerBlock: its non-trainable weights go last
            out.extend(held); held = []; block = None
        if top.startswith("conformer_"):
            block = top
            if name.endswith("/moving_mean") or name.endswith("/moving_variance"):
                held.append((name, _keras_shape(name, tuple(shape))))
                continue
        out.append((name, _keras_shape(name, tuple(shape))))
    out.extend(held)
    return out


def to_keras_list(weights: Dict[str, np.ndarray], entries: Sequence[Entry]) -> List[np.ndarray]:
    """Library weights (`Model.get_weights()`) -> the list `keras_model.set_weights` takes."""
    return [np.asarray(weights[n], np.float32).reshape(ks) for n, ks in keras_weight_order(entries)]


def from_keras_list(arrays: Sequence[np.ndarray], entries: Sequence[Entry]) -> Dict[str, np.ndarray]:
    """`keras_model.get_weights()` -> library weights (`Model.set_weights`)."""
    order = keras_weight_order(entries)
    if len(arrays) != len(order):
        raise ValueError(f"expected {len(order)} arrays (Keras model.get_weights()), got {len(arrays)}")
    shapes = {n: tuple(s) for n, s in entries}
    out = {}
    for a, (n, ks) in zip(arrays, order):
        a = np.asarray(a, np.float32)
        if tuple(a.shape) != tuple(ks):
            raise ValueError(f"{n}: Keras shape {ks} expected, got {tuple(a.shape)}")
        out[n] = a.reshape(shapes[n])
    return out


def save_keras_npz(path: str, weights: Dict[str, np.ndarray], entries: Sequence[Entry]):
    """Writes arr_0.. in Keras order: `model.set_weights([z[f"arr_{i}"] for i in range(len(z.files))])` loads it."""
    np.savez(path, *to_keras_list(weights, entries))


def load_keras_npz(path: str, entries: Sequence[Entry]) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return from_keras_list([z[f"arr_{i}"] for i in range(len(z.files))], entries)
