"""Keras `.h5` weight files for the Ishara model (reference: `model.save_weights("model.h5")`,
conv-hybrid-model.ipynb c9:10, and the matching `model.load_weights`).

The container is written and read by the HDF5 C library itself (`libhdf5.so`, present in this image under
/opt/conda/lib; bound with ctypes — there is no h5py here).  The LAYOUT is the one Keras 2 defines for
`save_weights(..., save_format="h5")` (`hdf5_format.save_weights_to_hdf5_group`):

    /                       attrs  layer_names  [n] fixed-length byte strings   (model.layers order)
                                   backend = b"tensorflow", keras_version = b"2.x"
    /<layer>                attrs  weight_names [k] fixed-length byte strings   (layer.weights order: trainable, then non-trainable)
    /<layer>/<weight name>  float32 dataset of the Keras shape; "/" in a weight name nests groups

`load_weights` (by_name=False) walks `layer_names` / `weight_names` IN ORDER and checks counts and shapes, so the order
and shapes are what make a file loadable; both come from `keras_interchange.keras_weight_order` (a Keras layer = the first
path component of the library's parameter names; the ConformerBlock's moving statistics trail the block).  Layers without
weights (Masking, Dropout, the TFOpLambda of `x + pe`) are not listed: Keras skips weightless layers on both sides when it
matches a file to a model.  Weight names are `<library name>:0`.

PARITY UNPINNED: no TensorFlow / Keras here and the reference holds no `.h5`, so that a file written here loads into the
reference's model — and the reverse — is not verified; what IS verified (tests/test_keras_h5.py) is that the HDF5 library
reads the file back with exactly this structure, bit-identical values, and that `load_weights_h5` rejects wrong shapes."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .keras_interchange import Entry, keras_weight_order

_LIB = None
hid_t = C.c_int64
hsize_t = C.c_uint64


def _lib():
    """libhdf5 with the handful of entry points used here; raises RuntimeError when the library cannot be loaded."""
    global _LIB
    if _LIB is not None:
        return _LIB
    cands = [os.environ.get("ISHARA_LIBHDF5", "")]
    cands += [os.path.join(p, n) for p in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/local/lib")
              for n in ("libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so")]
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    lib = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            break
        except OSError:
            continue
    if lib is None:
        raise RuntimeError("keras_h5: libhdf5 not found (set ISHARA_LIBHDF5); use keras_interchange.save_keras_npz instead")
    sig = {
        "H5open": (C.c_int, []), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (C.c_int, [hid_t]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
        "H5Sclose": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aget_type": (hid_t, [hid_t]),
        "H5Aget_space": (hid_t, [hid_t]), "H5Aclose": (C.c_int, [hid_t]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tset_strpad": (C.c_int, [hid_t, C.c_int]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tclose": (C.c_int, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (C.c_int, [hid_t, C.c_uint]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    if lib.H5open() < 0:
        raise RuntimeError("keras_h5: H5open failed")
    lib.H5Eset_auto2(0, None, None)            # errors are reported by return codes here, not printed by the library
    g = lambda sym: hid_t.in_dll(lib, sym).value
    lib.T_F32 = g("H5T_NATIVE_FLOAT_g"); lib.T_F32LE = g("H5T_IEEE_F32LE_g"); lib.T_S1 = g("H5T_C_S1_g")
    lib.P_LCPL = g("H5P_CLS_LINK_CREATE_ID_g")
    _LIB = lib
    return lib


def available() -> bool:
    try:
        _lib()
        return True
    except RuntimeError:
        return False


def _ck(v, what):
    if v < 0:
        raise IOError(f"keras_h5: {what} failed")
    return v


def _write_str_attr(lib, loc, name: str, values: Sequence[bytes], scalar: bool = False):
    """Fixed-length, null-padded byte strings — what h5py writes for a numpy 'S' array (Keras' layer_names / weight_names)."""
    n = max(1, max((len(v) for v in values), default=1))
    t = _ck(lib.H5Tcopy(lib.T_S1), "H5Tcopy"); lib.H5Tset_size(t, n); lib.H5Tset_strpad(t, 1)     # H5T_STR_NULLPAD
    if scalar:
        sp = _ck(lib.H5Screate(0), "H5Screate")                                                    # H5S_SCALAR
    else:
        dims = (hsize_t * 1)(len(values))
        sp = _ck(lib.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _ck(lib.H5Acreate2(loc, name.encode(), t, sp, 0, 0), f"H5Acreate2({name})")
    buf = b"".join(v.ljust(n, b"\0") for v in values) or b"\0"
    _ck(lib.H5Awrite(a, t, C.c_char_p(buf)), "H5Awrite")
    lib.H5Aclose(a); lib.H5Sclose(sp); lib.H5Tclose(t)


def _read_str_attr(lib, loc, name: str) -> List[bytes]:
    a = _ck(lib.H5Aopen(loc, name.encode(), 0), f"H5Aopen({name})")
    t = lib.H5Aget_type(a); sp = lib.H5Aget_space(a)
    n = lib.H5Tget_size(t)
    nd = lib.H5Sget_simple_extent_ndims(sp)
    cnt = 1
    if nd > 0:
        dims = (hsize_t * nd)()
        lib.H5Sget_simple_extent_dims(sp, dims, None)
        cnt = int(np.prod([dims[i] for i in range(nd)]))
    buf = C.create_string_buffer(max(1, n * cnt))
    mt = lib.H5Tcopy(lib.T_S1); lib.H5Tset_size(mt, n); lib.H5Tset_strpad(mt, 1)
    _ck(lib.H5Aread(a, mt, buf), "H5Aread")
    lib.H5Tclose(mt); lib.H5Tclose(t); lib.H5Sclose(sp); lib.H5Aclose(a)
    raw = buf.raw
    return [raw[i * n:(i + 1) * n].rstrip(b"\0") for i in range(cnt)]


def _layers(entries: Sequence[Entry]) -> List[Tuple[str, List[Tuple[str, Tuple[int, ...]]]]]:
    """[(Keras layer name, [(library weight name, Keras shape), ...])] in file order."""
    out: List[Tuple[str, List[Tuple[str, Tuple[int, ...]]]]] = []
    for name, ks in keras_weight_order(entries):
        top = name.split("/")[0]
        if not out or out[-1][0] != top:
            out.append((top, []))
        out[-1][1].append((name, ks))
    return out


def save_weights_h5(path: str, weights: Dict[str, np.ndarray], entries: Sequence[Entry], keras_version: str = "2.11.0"):
    """Library weights (`Model.get_weights()`) -> a Keras-2 `save_weights` HDF5 file.  `entries`: (name, shape) in library order."""
    lib = _lib()
    f = _ck(lib.H5Fcreate(path.encode(), 2, 0, 0), f"H5Fcreate({path})")                            # H5F_ACC_TRUNC
    lcpl = _ck(lib.H5Pcreate(lib.P_LCPL), "H5Pcreate"); lib.H5Pset_create_intermediate_group(lcpl, 1)
    try:
        layers = _layers(entries)
        _write_str_attr(lib, f, "layer_names", [ln.encode() for ln, _ in layers])
        _write_str_attr(lib, f, "backend", [b"tensorflow"], scalar=True)
        _write_str_attr(lib, f, "keras_version", [keras_version.encode()], scalar=True)
        for ln, ws in layers:
            g = _ck(lib.H5Gcreate2(f, ln.encode(), 0, 0, 0), f"H5Gcreate2({ln})")
            _write_str_attr(lib, g, "weight_names", [(n + ":0").encode() for n, _ in ws])
            for n, ks in ws:
                arr = np.ascontiguousarray(np.asarray(weights[n], np.float32).reshape(ks))
                dims = (hsize_t * max(1, arr.ndim))(*arr.shape) if arr.ndim else None
                sp = _ck(lib.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else lib.H5Screate(0), "dataspace")
                d = _ck(lib.H5Dcreate2(g, (n + ":0").encode(), lib.T_F32LE, sp, lcpl, 0, 0), f"H5Dcreate2({n})")
                _ck(lib.H5Dwrite(d, lib.T_F32, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), f"H5Dwrite({n})")
                lib.H5Dclose(d); lib.H5Sclose(sp)
            lib.H5Gclose(g)
    finally:
        lib.H5Pclose(lcpl)
        lib.H5Fclose(f)


def read_h5(path: str) -> List[Tuple[str, List[Tuple[str, np.ndarray]]]]:
    """[(layer name, [(weight name, float32 array), ...])] in the order the file lists them (`layer_names` / `weight_names`)."""
    lib = _lib()
    f = _ck(lib.H5Fopen(path.encode(), 0, 0), f"H5Fopen({path})")                                   # H5F_ACC_RDONLY
    out = []
    try:
        for ln in _read_str_attr(lib, f, "layer_names"):
            g = _ck(lib.H5Gopen2(f, ln, 0), f"H5Gopen2({ln!r})")
            ws = []
            names = _read_str_attr(lib, g, "weight_names") if lib.H5Aexists(g, b"weight_names") > 0 else []
            for wn in names:
                d = _ck(lib.H5Dopen2(g, wn, 0), f"H5Dopen2({wn!r})")
                sp = lib.H5Dget_space(d)
                nd = lib.H5Sget_simple_extent_ndims(sp)
                shape: Tuple[int, ...] = ()
                if nd > 0:
                    dims = (hsize_t * nd)()
                    lib.H5Sget_simple_extent_dims(sp, dims, None)
                    shape = tuple(int(dims[i]) for i in range(nd))
                arr = np.empty(shape, np.float32)
                _ck(lib.H5Dread(d, lib.T_F32, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), f"H5Dread({wn!r})")
                lib.H5Sclose(sp); lib.H5Dclose(d)
                ws.append((wn.decode(), arr))
            lib.H5Gclose(g)
            out.append((ln.decode(), ws))
    finally:
        lib.H5Fclose(f)
    return out


def load_weights_h5(path: str, entries: Sequence[Entry]) -> Dict[str, np.ndarray]:
    """A Keras-2 weights file -> library weights (`Model.set_weights`), matched the way `load_weights(by_name=False)` matches:
    layers with weights in order, weights within a layer in order, counts and shapes checked (names are not)."""
    want = _layers(entries)
    have = [(ln, ws) for ln, ws in read_h5(path) if ws]
    if len(have) != len(want):
        raise ValueError(f"{path}: {len(have)} layers with weights, the model has {len(want)}")
    shapes = {n: tuple(s) for n, s in entries}
    out: Dict[str, np.ndarray] = {}
    for (ln, ws), (wl, wws) in zip(have, want):
        if len(ws) != len(wws):
            raise ValueError(f"{path}: layer {ln!r} holds {len(ws)} weights, the model's {wl!r} expects {len(wws)}")
        for (fn, arr), (n, ks) in zip(ws, wws):
            if tuple(arr.shape) != tuple(ks):
                raise ValueError(f"{path}: {ln}/{fn} has shape {tuple(arr.shape)}, {n} expects Keras shape {tuple(ks)}")
            out[n] = arr.reshape(shapes[n])
    return out
