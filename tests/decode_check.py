"""Greedy-decode parity bookkeeping shared by the GPU parity tests (north_star: "identical greedy decode indices";
reference: decode_phrase, conv-hybrid-model.ipynb c8:4-12, and the TFLite wrapper's one_hot output, c13:17-24).

A 16-bit run cannot reproduce the argmax of a frame whose two best logits are closer than its own logit error, so the
comparison is made where the oracle's decision is resolved — and it COUNTS what it compared, so that an empty comparison
fails instead of passing (`all([])` is True):

  * per FRAME: argmax of the HIP logits == argmax of the oracle logits on every frame whose oracle top-2 margin exceeds
    `2 * err` (err = the observed max-abs logit error of the run); asserts zero mismatches, at least `min_frac` of all
    frames compared, and at least one frame;
  * per CLIP: `decode_phrase` indices identical on every clip all of whose frames are resolved in that sense;
    `require_clips` makes "no such clip" a failure (the confident-logits cases).
"""
import numpy as np


def frame_margins(ref_logits):
    top2 = np.sort(ref_logits, axis=-1)[..., -2:]
    return top2[..., 1] - top2[..., 0]


def check_decode_parity(ref_logits, got_logits, got_decodes, decode_phrase, err=None, min_frac=0.0, require_clips=0, what=""):
    """ref_logits / got_logits [B,T,C]; got_decodes: the HIP decoder's index arrays per clip.  Returns the counts (logged by callers)."""
    ref_logits, got_logits = np.asarray(ref_logits), np.asarray(got_logits)
    if ref_logits.ndim == 2:
        ref_logits, got_logits, got_decodes = ref_logits[None], got_logits[None], [got_decodes]
    B, T, _ = ref_logits.shape
    if err is None:
        err = float(np.abs(got_logits - ref_logits).max())
    margin = frame_margins(ref_logits)
    clear = margin > 2 * max(err, 1e-7)
    n_clear = int(clear.sum())
    mism = int((np.argmax(got_logits, -1)[clear] != np.argmax(ref_logits, -1)[clear]).sum())
    clips = [b for b in range(B) if clear[b].all()]
    clip_bad = [b for b in clips if not np.array_equal(np.asarray(got_decodes[b]), decode_phrase(ref_logits[b]))]
    rec = dict(frames=B * T, frames_compared=n_clear, frame_mismatches=mism, frac_compared=n_clear / (B * T),
               clips=B, clips_compared=len(clips), clip_mismatches=len(clip_bad),
               nonempty_decodes_compared=int(sum(len(decode_phrase(ref_logits[b])) > 0 for b in clips)), logit_err=err)
    assert n_clear >= 1, f"{what}: no frame's top-2 margin exceeds 2x the logit error {err:.3e} — nothing was compared"
    assert mism == 0, f"{what}: {mism} of {n_clear} resolved frames decode to another class than the oracle's"
    assert n_clear >= min_frac * B * T, f"{what}: only {n_clear}/{B * T} frames resolved (< {min_frac:.0%})"
    assert not clip_bad, f"{what}: decode_phrase differs from the oracle's on fully resolved clips {clip_bad}"
    assert len(clips) >= require_clips, f"{what}: {len(clips)} fully resolved clips, {require_clips} required"
    return rec
