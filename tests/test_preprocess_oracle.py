"""CPU: host-side logic of the TFLite-shaped wrapper restatement (oracle/preprocess_oracle.py)."""
import numpy as np

from oracle import preprocess_oracle as PO


def test_column_order_and_shapes():
    x = np.arange(5 * 276, dtype=np.float32).reshape(5, 276)
    p = PO.split_parts(x)
    assert {k: v.shape for k, v in p.items()} == {"lip": (5, 40, 3), "rhand": (5, 21, 3), "lhand": (5, 21, 3), "rpose": (5, 5, 3), "lpose": (5, 5, 3)}
    assert p["rhand"][0, 0].tolist() == [0.0, 92.0, 184.0]          # x_right_hand_0, y_..., z_... (c1:22-26)
    assert p["lip"][0, 0, 0] == 52.0 and p["lpose"][0, 0, 0] == 42.0 and p["rpose"][0, 0, 0] == 47.0
    assert PO.preprocess(x, 176).shape == (176, 276)


def test_frame_filter_keeps_hands_or_even_frames():
    x = np.full((6, 276), np.nan, np.float32)
    x[:, 52:92] = 1.0                                                # lips only: no hands anywhere
    assert PO.frame_mask(x).tolist() == [True, False, True, False, True, False]
    x[3, 0] = 0.5                                                    # a hand coordinate on an odd frame
    assert PO.frame_mask(x).tolist() == [True, False, True, True, True, False]


def test_resize_pad():
    a = np.arange(4, dtype=np.float32)[:, None, None] * np.ones((1, 2, 3), np.float32)
    p = PO.resize_pad(a, 6)
    assert p.shape == (6, 2, 3) and np.isnan(p[4:]).all() and np.array_equal(p[:4], a)
    r = PO.resize_pad(np.arange(8, dtype=np.float32)[:, None, None] * np.ones((1, 1, 3), np.float32), 4)[:, 0, 0]
    assert np.allclose(r, [0.5, 2.5, 4.5, 6.5])                      # half-pixel centres, scale 2
    empty = PO.preprocess(np.zeros((0, 276), np.float32), 8)
    assert empty.shape == (8, 276) and (empty == 0).all()
