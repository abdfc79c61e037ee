"""TFRecord front end of the training notebook: the step in front of the hot path.

`Test Notebooks/conv-hybrid-model.ipynb` c4:1-40 feeds `model.fit` from
`tf.data.TFRecordDataset(files).map(decode_fn).map(pre_process_fn).batch(B)`.  TensorFlow is not available here (and is
not needed for this): the container and the `tf.train.Example` wire format are small public formats, restated below.

* TFRecord container: per record `u64 length | u32 masked_crc32c(length) | data | u32 masked_crc32c(data)`, little endian,
  mask(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32 (tensorflow/core/lib/io/record_writer.h, hash/crc32c.h).
* `tf.train.Example` (tensorflow/core/example/example.proto, feature.proto):
  Example{1: Features}; Features{1: map<string, Feature>} (map entry {1: key, 2: value});
  Feature oneof {1: BytesList, 2: FloatList, 3: Int64List}; *List{1: repeated value} (floats / int64 packed or not).
* `decode_fn` (c4:1-21): VarLenFeature float `lip|rhand|lhand|rpose|lpose` reshaped to (-1, 40|21|21|5|5, 3), int64 `phrase`.
* `pre_process_fn` (c4:23-25) = (`pre_process1` (c3:103-115), phrase padded to 64 with 59).

Host-side numpy, like the reference's tf.data pipeline (CPU threads feeding the device step); the GPU hot path starts
at the batch these functions produce.  PARITY: the CRC-32C check value and an independent protobuf-library encoding pin
the formats (tests/test_tfrecord.py); `pre_process1` is unpinned against TF (tf.image.resize unavailable) and follows the
same restatement as the on-device `ishara_preprocess`."""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

MAX_PHRASE_LENGTH = 64      # c1:33
PAD_TOKEN_IDX = 59          # c1:5
PARTS = (("lip", 40), ("rhand", 21), ("lhand", 21), ("rpose", 5), ("lpose", 5))     # c4:3-8, concat order c3:111

# ------------------------------------------------------------------ CRC-32C (Castagnoli), table driven
_POLY = 0x82F63B78
_TABLE = np.zeros(256, dtype=np.uint32)
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE[_i] = _c
_TABLE_L = [int(v) for v in _TABLE]


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    t = _TABLE_L
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------ TFRecord container
def write_records(path: str, records: Iterable[bytes]) -> int:
    n = 0
    with open(path, "wb") as f:
        for r in records:
            hdr = struct.pack("<Q", len(r))
            f.write(hdr)
            f.write(struct.pack("<I", masked_crc32c(hdr)))
            f.write(r)
            f.write(struct.pack("<I", masked_crc32c(r)))
            n += 1
    return n


def read_records(path: str, verify: bool = True) -> Iterator[bytes]:
    """Yields the payload of every record; raises ValueError on a truncated file, a length-CRC mismatch or (verify) a
    data-CRC mismatch."""
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) != 8:
                raise ValueError(f"{path}: truncated record header")
            (ln,) = struct.unpack("<Q", hdr)
            crc_h = f.read(4)
            if len(crc_h) != 4:
                raise ValueError(f"{path}: truncated record")
            if struct.unpack("<I", crc_h)[0] != masked_crc32c(hdr):       # always: a corrupt length must not be trusted
                raise ValueError(f"{path}: length CRC mismatch")
            data = f.read(ln)
            crc_d = f.read(4)
            if len(data) != ln or len(crc_d) != 4:
                raise ValueError(f"{path}: truncated record")
            if verify and struct.unpack("<I", crc_d)[0] != masked_crc32c(data):
                raise ValueError(f"{path}: data CRC mismatch")
            yield data


# ------------------------------------------------------------------ protobuf wire format (the subset Example uses)
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    v, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v: int) -> bytes:
    v &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; value = int (varint, fixed) or bytes (length-delimited)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            if len(v) != ln:
                raise ValueError("truncated length-delimited field")
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _parse_feature(buf: bytes) -> np.ndarray:
    for fn, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fn == 2:        # FloatList
            chunks: List[np.ndarray] = []
            for f2, w2, v2 in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 2: chunks.append(np.frombuffer(v2, dtype="<f4"))          # packed
                elif w2 == 5: chunks.append(np.frombuffer(v2, dtype="<f4"))        # one unpacked value
            return np.concatenate(chunks).astype(np.float32) if chunks else np.zeros(0, np.float32)
        if fn == 3:        # Int64List
            vals: List[int] = []
            for f2, w2, v2 in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 2:
                    p = 0
                    while p < len(v2):
                        x, p = _varint(v2, p)
                        vals.append(x)
                elif w2 == 0:
                    vals.append(v2)
            a = np.array(vals, dtype=np.uint64).astype(np.int64) if vals else np.zeros(0, np.int64)   # two's complement
            return a
        if fn == 1:        # BytesList -> object array of bytes
            return np.array([v2 for f2, w2, v2 in _fields(v) if f2 == 1 and w2 == 2], dtype=object)
    return np.zeros(0, np.float32)


def parse_example(record: bytes) -> Dict[str, np.ndarray]:
    """tf.io.parse_single_example with every feature VarLen: name -> 1-D array (float32 / int64 / bytes objects)."""
    out: Dict[str, np.ndarray] = {}
    for fn, wt, feats in _fields(record):
        if fn != 1 or wt != 2:
            continue
        for f2, w2, entry in _fields(feats):
            if f2 != 1 or w2 != 2:
                continue
            key, val = None, None
            for f3, w3, v3 in _fields(entry):
                if f3 == 1 and w3 == 2: key = v3.decode("utf-8")
                elif f3 == 2 and w3 == 2: val = v3
            if key is not None:
                out[key] = _parse_feature(val if val is not None else b"")
    return out


def _ld(fn: int, payload: bytes) -> bytes:
    return _put_varint((fn << 3) | 2) + _put_varint(len(payload)) + payload


def encode_example(features: Dict[str, np.ndarray]) -> bytes:
    """Serialises name -> array as a tf.train.Example (float arrays -> FloatList, integer arrays -> Int64List, both packed)."""
    body = b""
    for name in sorted(features):
        a = np.asarray(features[name])
        if a.dtype.kind == "f":
            feat = _ld(2, _ld(1, a.astype("<f4").ravel().tobytes()))
        elif a.dtype.kind in "iu":
            feat = _ld(3, _ld(1, b"".join(_put_varint(int(v)) for v in a.ravel())))
        else:
            raise TypeError(f"feature {name}: unsupported dtype {a.dtype}")
        body += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feat))
    return _ld(1, body)


# ------------------------------------------------------------------ the notebook's map functions
def decode_fn(record_bytes: bytes):
    """c4:1-21 -> (lip [n,40,3], rhand [n,21,3], lhand [n,21,3], rpose [n,5,3], lpose [n,5,3], phrase int64 [l])."""
    x = parse_example(record_bytes)
    parts = []
    for name, cnt in PARTS:
        a = x.get(name, np.zeros(0, np.float32)).astype(np.float32)
        if a.size % (cnt * 3) != 0:
            raise ValueError(f"feature {name}: {a.size} values do not reshape to (-1, {cnt}, 3)")
        parts.append(a.reshape(-1, cnt, 3))
    return (*parts, x.get("phrase", np.zeros(0, np.int64)).astype(np.int64))


def resize_pad(a: np.ndarray, T: int) -> np.ndarray:
    """c3:1-7.  a [n, c, 3]: NaN-pad to T frames, else bilinear resize over the frame axis (tf.image.resize default:
    half-pixel centres, no antialias)."""
    n = a.shape[0]
    a = a.astype(np.float32)
    if n < T:
        return np.concatenate([a, np.full((T - n,) + a.shape[1:], np.nan, dtype=np.float32)], axis=0)
    src = (np.arange(T, dtype=np.float32) + np.float32(0.5)) * np.float32(n / T) - np.float32(0.5)
    src = np.clip(src, 0.0, None)
    i0 = np.minimum(np.floor(src).astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    w = (src - i0.astype(np.float32)).astype(np.float32)[:, None, None]
    return (a[i0] * (np.float32(1.0) - w) + a[i1] * w).astype(np.float32)


def pre_process1(lip, rhand, lhand, rpose, lpose, T: int, stats: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]] = None) -> np.ndarray:
    """c3:103-115: per part (resize_pad - mean) / std, concat on the landmark axis, flatten to [T, 92*3], NaN -> 0.
    `stats`: part -> (mean [cnt,3], std [cnt,3]) (the notebook's LIPM/LIPS ... .npy files); default (0, 1)."""
    cols = []
    for (name, cnt), a in zip(PARTS, (lip, rhand, lhand, rpose, lpose)):
        mean, std = (stats[name] if stats else (np.zeros((cnt, 3), np.float32), np.ones((cnt, 3), np.float32)))
        cols.append((resize_pad(np.asarray(a, np.float32), T) - mean) / std)
    x = np.concatenate(cols, axis=1)
    x = x.reshape(x.shape[0], x.shape[1] * x.shape[2])
    return np.where(np.isnan(x), 0.0, x).astype(np.float32)


def pre_process_fn(lip, rhand, lhand, rpose, lpose, phrase, T: int = 384, stats=None):
    """c4:23-25 -> (x float32 [T, 276], phrase int64 [64] padded with 59)."""
    phrase = np.asarray(phrase, np.int64)
    if phrase.shape[0] > MAX_PHRASE_LENGTH:
        raise ValueError(f"phrase of {phrase.shape[0]} tokens exceeds MAX_PHRASE_LENGTH={MAX_PHRASE_LENGTH} (tf.pad would fail too)")
    y = np.full(MAX_PHRASE_LENGTH, PAD_TOKEN_IDX, dtype=np.int64)
    y[:phrase.shape[0]] = phrase
    return pre_process1(lip, rhand, lhand, rpose, lpose, T, stats), y


class TFRecordDataset:
    """Re-iterable `(x [B,T,276] float32, y [B,64] int64)` batches from TFRecord files: the object `Model.fit` takes in
    place of c4:33-44's `TFRecordDataset(...).map(decode_fn).map(pre_process_fn).batch(B)`.  `shuffle` > 0 keeps a
    buffer of that many decoded records and draws from it with `numpy.random.default_rng(seed + epoch)` (the
    notebook's `.shuffle(5000)`); the last partial batch is kept, like tf.data's default `drop_remainder=False`."""

    def __init__(self, files: Sequence[str], batch_size: int, T: int = 384, stats=None, shuffle: int = 0, seed: int = 0,
                 verify_crc: bool = True):
        self.files, self.batch_size, self.T, self.stats = list(files), int(batch_size), int(T), stats
        self.shuffle, self.seed, self.verify_crc, self._epoch = int(shuffle), int(seed), verify_crc, 0

    def _samples(self):
        for path in self.files:
            for rec in read_records(path, self.verify_crc):
                yield pre_process_fn(*decode_fn(rec), T=self.T, stats=self.stats)

    def __iter__(self):
        it = self._samples()
        if self.shuffle > 0:
            rng = np.random.default_rng(self.seed + self._epoch)
            it = self._shuffled(it, rng)
        self._epoch += 1
        xs, ys = [], []
        for x, y in it:
            xs.append(x); ys.append(y)
            if len(xs) == self.batch_size:
                yield np.stack(xs), np.stack(ys)
                xs, ys = [], []
        if xs:
            yield np.stack(xs), np.stack(ys)

    def _shuffled(self, it, rng):
        buf = []
        for s in it:
            if len(buf) < self.shuffle:
                buf.append(s)
                continue
            i = int(rng.integers(0, len(buf)))
            out, buf[i] = buf[i], s
            yield out
        order = rng.permutation(len(buf))
        for i in order:
            yield buf[int(i)]
