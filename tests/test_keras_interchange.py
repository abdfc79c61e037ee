"""Keras get_weights()/set_weights() interchange order (ishara_amd/keras_interchange.py).  The ordering itself cannot be
verified without TensorFlow (parity unpinned, see the module header); checked here: round trip, Keras shapes, and that
the ConformerBlock's non-trainable weights trail the block.  Entries come from the oracle's param_specs, which lists the
same (name, shape) pairs as `Model.entries` (asserted on the GPU in test_model_gpu.py::test_keras_interchange_matches_model)."""
import numpy as np

from ishara_amd import keras_interchange as K
from oracle import ishara_oracle as O


def _entries(**kw):
    cfg = O.Config(**kw)
    return [(n, tuple(s)) for n, s, _, _ in O.param_specs(cfg)], cfg


def test_order_and_shapes():
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    order = K.keras_weight_order(ent)
    names = [n for n, _ in order]
    assert sorted(names) == sorted(n for n, _ in ent) and len(set(names)) == len(names)
    # functional layers keep trainable-then-moving order in place
    i = names.index("stem_bn/gamma")
    assert names[i:i + 4] == ["stem_bn/gamma", "stem_bn/beta", "stem_bn/moving_mean", "stem_bn/moving_variance"]
    # the ConformerBlock is one Keras layer: its moving statistics come after all its trainable weights
    last_train = names.index("conformer_0/layer_norm2/beta")
    assert names[last_train + 1: last_train + 3] == ["conformer_0/conv/batch_norm/moving_mean", "conformer_0/conv/batch_norm/moving_variance"]
    assert names[last_train + 3] == "top_conv/kernel"
    shapes = dict(order)
    assert shapes["convsqueeze_0_1_dwconv/depthwise_kernel"] == (11, 128, 1)
    assert shapes["conformer_0/conv/depthwise_conv/kernel"] == (15, 1, 64)
    assert shapes["convsqueeze_0_1_eca/kernel"] == (5, 1, 1)
    assert shapes["squeezeformer_0/conv/conv1/kernel"] == (1, 64, 128)
    assert shapes["conformer_0/conv/pointwise_conv2/kernel"] == (1, 64, 64)
    assert shapes["squeezeformer_0/mha/qkv/kernel"] == (64, 192) and shapes["classifier/bias"] == (60,)


def test_round_trip(tmp_path):
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, input_shape=(176, 276))
    W = O.init_params(cfg, 3)
    lst = K.to_keras_list(W, ent)
    assert sum(a.size for a in lst) == sum(int(np.prod(s)) for _, s in ent)
    back = K.from_keras_list(lst, ent)
    assert all(np.array_equal(back[n], W[n]) for n, _ in ent)
    p = str(tmp_path / "w.npz")
    K.save_keras_npz(p, W, ent)
    back2 = K.load_keras_npz(p, ent)
    assert all(np.array_equal(back2[n], W[n]) for n, _ in ent)
    try:
        K.from_keras_list(lst[:-1], ent)
        assert False
    except ValueError:
        pass
