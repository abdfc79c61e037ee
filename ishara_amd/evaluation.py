"""Evaluation side of the hot path: the normalised-Levenshtein score of the
inference notebook cell (conv-hybrid-model.ipynb c18:1-15) and the per-epoch transcription report
(CallbackEval, c9:1-29).  Host-side string work; the logits -> index decode it consumes is the HIP
greedy decoder (`Model.decode_batch`)."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

PAD_TOKEN = "^"        # c1:4
PAD_TOKEN_IDX = 59     # c1:5


def levenshtein(a: Sequence, b: Sequence) -> int:
    """Edit distance with unit insert / delete / substitute costs — `Levenshtein.distance` (c18:1, 9)."""
    if len(a) < len(b):
        a, b = b, a
    if len(b) == 0:
        return len(a)
    prev = np.arange(len(b) + 1)
    bb = np.asarray(list(b), dtype=object)
    for i, ca in enumerate(a, 1):
        cur = np.empty_like(prev)
        cur[0] = i
        sub = prev[:-1] + (bb != ca)
        dele = prev[1:] + 1
        best = np.minimum(sub, dele)
        # insertions are a running minimum along the row
        run = i
        for j in range(1, len(b) + 1):
            run = min(run + 1, best[j - 1])
            cur[j] = run
        prev = cur
    return int(prev[-1])


def normalized_score(prediction: str, target: str) -> float:
    """(len(target) - distance(prediction, target)) / len(target) — c18:9."""
    return (len(target) - levenshtein(prediction, target)) / len(target)


def mean_score(predictions: Iterable[str], targets: Iterable[str]) -> float:
    """np.sum(scores) / len(scores) — c18:13-15."""
    scores = [normalized_score(p, t) for p, t in zip(predictions, targets)]
    return float(np.sum(scores) / len(scores)) if scores else 0.0


def make_num_to_char(char_to_num: Dict[str, int]) -> Dict[int, str]:
    """num_to_char with the pad token added at 59 — c1:3-9."""
    m = dict(char_to_num)
    m[PAD_TOKEN] = PAD_TOKEN_IDX
    return {j: i for i, j in m.items()}


def num_to_char_fn(y, num_to_char: Dict[int, str]) -> List[str]:
    """[num_to_char.get(x, "") for x in y] — c8:1-2."""
    return [num_to_char.get(int(x), "") for x in y]


class CallbackEval:
    """Displays a batch of outputs after every epoch (c9:1-29): saves the weights, decodes every batch of
    `dataset` greedily and prints target / prediction pairs.  `model` is an `ishara_amd.Model`."""

    def __init__(self, dataset: Iterable, num_to_char: Dict[int, str], n_show: int = 32,
                 weights_path: Optional[str] = "model.npz", printer: Callable[[str], None] = print):
        self.dataset = dataset
        self.num_to_char = num_to_char
        self.n_show = n_show
        self.weights_path = weights_path
        self.printer = printer
        self.model = None
        self.last_score: Optional[float] = None

    def set_model(self, model):
        self.model = model

    def on_epoch_begin(self, epoch: int, logs=None):
        pass

    def on_epoch_end(self, epoch: int, logs=None):
        model = self.model
        if self.weights_path:
            model.save_weights(self.weights_path)                                   # c9:10
        predictions, targets = [], []
        for X, y in self.dataset:                                                   # c9:13-21
            logits = model(X, training=False)
            for idx in model.decode_batch(logits):
                predictions.append("".join(num_to_char_fn(idx, self.num_to_char)))
            for label in np.asarray(y.cpu() if hasattr(y, "cpu") else y):
                targets.append("".join(num_to_char_fn(label, self.num_to_char)))    # includes the pad characters, as c9:20 does
        self.printer("-" * 100)
        for i in range(min(self.n_show, len(predictions))):                         # c9:24-27
            self.printer(f"Target    : {targets[i]}")
            self.printer(f"Prediction: {predictions[i]}, len: {len(predictions[i])}")
            self.printer("-" * 100)
        stripped = [t.replace(PAD_TOKEN, "") for t in targets]
        pairs = [(p, t) for p, t in zip(predictions, stripped) if len(t) > 0]
        self.last_score = mean_score([p for p, _ in pairs], [t for _, t in pairs])
        if logs is not None:
            logs["val_levenshtein"] = self.last_score
