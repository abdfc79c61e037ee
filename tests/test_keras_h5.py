"""Keras `.h5` weight files (ishara_amd/keras_h5.py; reference: model.save_weights("model.h5"), conv-hybrid-model.ipynb c9:10).
The container is written and read by the HDF5 C library of the image; checked here: the file has the Keras-2 save_weights
structure (layer_names / weight_names attributes, one group per layer, nested datasets of the Keras shapes), the values come
back bit-identical, and a file whose shapes or counts do not fit the model is rejected.  That Keras itself loads the file is
UNPINNED (no TensorFlow here; the reference holds no .h5)."""
import numpy as np
import pytest

from ishara_amd import keras_h5 as H
from ishara_amd import keras_interchange as K
from oracle import ishara_oracle as O

pytestmark = pytest.mark.skipif(not H.available(), reason="libhdf5 is not loadable in this environment")


def _entries(**kw):
    cfg = O.Config(**kw)
    return [(n, tuple(s)) for n, s, _, _ in O.param_specs(cfg)], cfg


def test_round_trip_and_structure(tmp_path):
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    W = O.init_params(cfg, 5)
    p = str(tmp_path / "model.h5")
    H.save_weights_h5(p, W, ent)
    with open(p, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"                       # the HDF5 signature
    layers = H.read_h5(p)
    names = [ln for ln, _ in layers]
    assert names[0] == "stem_conv" and names[1] == "stem_bn" and names[-1] == "classifier" and len(set(names)) == len(names)
    flat = [(ln, wn, a) for ln, ws in layers for wn, a in ws]
    order = K.keras_weight_order(ent)
    assert [wn for _, wn, _ in flat] == [n + ":0" for n, _ in order]      # model.get_weights() order
    assert all(tuple(a.shape) == tuple(ks) and a.dtype == np.float32 for (_, _, a), (_, ks) in zip(flat, order))
    assert all(wn.split("/")[0] == ln for ln, wn, _ in flat)              # a weight sits in its layer's group
    bn = dict(layers)["stem_bn"]
    assert [wn for wn, _ in bn] == ["stem_bn/gamma:0", "stem_bn/beta:0", "stem_bn/moving_mean:0", "stem_bn/moving_variance:0"]
    conf = [wn for wn, _ in dict(layers)["conformer_0"]]
    assert conf[-2:] == ["conformer_0/conv/batch_norm/moving_mean:0", "conformer_0/conv/batch_norm/moving_variance:0"]
    back = H.load_weights_h5(p, ent)
    assert set(back) == {n for n, _ in ent}
    assert all(np.array_equal(back[n], W[n]) and back[n].shape == tuple(s) for n, s in ent)


def test_same_arrays_as_the_npz_interchange(tmp_path):
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, input_shape=(176, 276))
    W = O.init_params(cfg, 7)
    p = str(tmp_path / "w.h5")
    H.save_weights_h5(p, W, ent)
    lst = K.to_keras_list(W, ent)
    got = [a for _, ws in H.read_h5(p) for _, a in ws]
    assert len(got) == len(lst) and all(np.array_equal(a, b) for a, b in zip(got, lst))


def test_mismatched_model_is_rejected(tmp_path):
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    W = O.init_params(cfg, 1)
    p = str(tmp_path / "w.h5")
    H.save_weights_h5(p, W, ent)
    ent2, _ = _entries(dim=128, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    with pytest.raises(ValueError):
        H.load_weights_h5(p, ent2)
    ent3, _ = _entries(dim=64, num_conv_squeeze_blocks=2, num_conv_conform_blocks=1, input_shape=(176, 276))
    with pytest.raises(ValueError):
        H.load_weights_h5(p, ent3)
    with pytest.raises(IOError):
        H.read_h5(str(tmp_path / "missing.h5"))


def test_h5dump_sees_the_keras_layout(tmp_path):
    """An independent reader (the HDF5 distribution's own `h5dump`, when the image has it): /<layer>/<layer>/<weight>:0 datasets of
    IEEE f32, null-padded fixed-length string attributes — the nesting a file written by Keras through h5py shows."""
    import shutil, subprocess
    tool = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if __import__("os").path.exists("/opt/conda/bin/h5dump") else None)
    if tool is None:
        pytest.skip("no h5dump in this image")
    ent, cfg = _entries(dim=64, num_conv_squeeze_blocks=1, num_conv_conform_blocks=1, input_shape=(176, 276))
    p = str(tmp_path / "m.h5")
    H.save_weights_h5(p, O.init_params(cfg, 2), ent)
    out = subprocess.run([tool, "-H", p], capture_output=True, text=True, timeout=60).stdout
    assert out.count("DATASET") == len(ent) and "H5T_IEEE_F32LE" in out and "STRPAD H5T_STR_NULLPAD" in out
    assert 'GROUP "stem_bn" {' in out and 'DATASET "moving_variance:0"' in out and 'ATTRIBUTE "layer_names"' in out
    data = subprocess.run([tool, "-d", "/classifier/classifier/bias:0", p], capture_output=True, text=True, timeout=60).stdout
    assert "DATASPACE  SIMPLE { ( 60 ) / ( 60 ) }" in data
