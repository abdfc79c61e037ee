"""GPU: device preprocessing kernel and the TFLite-shaped wrapper against the numpy restatement
(oracle/preprocess_oracle.py) and the oracle model; hipGraph replay equals eager execution."""
import ctypes as C

import numpy as np
import pytest
import torch

from ishara_amd import _lib, get_model
from ishara_amd.tflite_model import TFLiteModel

pytestmark = pytest.mark.gpu


def _clip(n, seed, nan_hands=0.5):
    g = np.random.default_rng(seed)
    x = g.standard_normal((n, 276)).astype(np.float32)
    for f in range(n):                       # hands are often missing in real clips: whole hand blocks NaN
        if g.random() < nan_hands:
            for a in range(3):
                x[f, a * 92: a * 92 + 42] = np.nan
        if g.random() < 0.1:
            x[f, g.integers(0, 276, 5)] = np.nan
    return x


def _stats(seed=3):
    from oracle import preprocess_oracle as PO
    g = np.random.default_rng(seed)
    return {n: (0.1 * g.standard_normal((c, 3)).astype(np.float32), (0.5 + g.random((c, 3))).astype(np.float32)) for n, c in PO.PARTS}


@pytest.mark.parametrize("n", [0, 1, 7, 100, 176, 177, 300, 613])
def test_preprocess_kernel_matches_oracle(lib, n):
    from oracle import preprocess_oracle as PO
    T, maxf = 176, 1024
    stats = _stats()
    x = _clip(n, n)
    ref = PO.preprocess(x, T, stats)
    raw = torch.zeros(maxf, 276, device="cuda")
    if n:
        raw[:n] = torch.from_numpy(x).cuda()
    nd = torch.tensor([n], dtype=torch.int32, device="cuda")
    mean = torch.from_numpy(np.concatenate([stats[p][0].reshape(-1) for p, _ in PO.PARTS])).cuda()
    std = torch.from_numpy(np.concatenate([stats[p][1].reshape(-1) for p, _ in PO.PARTS])).cuda()
    out = torch.empty(T, 276, device="cuda")
    _lib.check(lib.ishara_preprocess(_lib.ptr(raw), _lib.ptr(nd), maxf, _lib.ptr(mean), _lib.ptr(std), _lib.ptr(out), T,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-5 * (1 + np.abs(ref).max()), f"n={n}"


@pytest.mark.parametrize("use_graph", [False, True])
def test_tflite_wrapper_end_to_end(use_graph, monkeypatch):
    from oracle import ishara_oracle as O
    from oracle import preprocess_oracle as PO
    kw = dict(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    monkeypatch.setenv("ISHARA_WS_GUARD", "1")       # guard zones between all workspace buffers, checked once at the end
    model = get_model(**kw, dtype="f32", max_batch=1, seed=5)
    assert model._lib.ishara_workspace_plan_check(model._h) > 0
    stats = _stats()
    tfl = TFLiteModel(model, stats=stats, max_frames=512, use_graph=use_graph)
    ocfg = O.Config(**kw)
    P = O.to_torch(model.get_weights(), torch.float64, requires_grad=False)
    for n in (0, 40, 250):
        x = _clip(n, 100 + n)
        out = tfl.get_signature_runner("serving_default")(inputs=x)["outputs"]
        xin = PO.preprocess(x, 176, stats)
        with torch.no_grad():
            logits, _ = O.forward(P, torch.from_numpy(xin)[None].double(), ocfg, training=False)
        lg = logits[0].numpy()
        want = O.tflite_postprocess(O.decode_phrase(lg))
        top2 = np.sort(lg, axis=1)[:, -2:]
        assert out.shape[1] == 59 and out.dtype == np.float32
        if (top2[:, 1] - top2[:, 0]).min() > 1e-3:            # no near-tie frame: identical indices
            assert out.shape == want.shape and np.array_equal(out, want), f"n={n}"
        assert np.abs(tfl._logits[0].cpu().numpy() - lg).max() <= 1e-4
    # no kernel of the forward pass (eager or replayed from the hipGraph) wrote outside its workspace buffer
    _lib.check(model._lib.ishara_workspace_guard_check(model._h), "workspace guard")
