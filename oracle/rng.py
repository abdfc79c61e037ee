"""Counter-based dropout RNG — CPU restatement of ishara_amd/csrc/rng.h.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the
product path.

The reference draws its dropout masks from TensorFlow's stateful RNG
(`tf.keras.layers.Dropout`, conv-hybrid-model.ipynb c5:83,98,164,183), which
is not reproducible outside TF.  The build therefore defines its own
counter-based generator so that the HIP kernels and this oracle can produce
bit-identical masks for any (seed, site, row, col):

    key_site = lowbias32(seed ^ (site * 0x9E3779B9))
    key_row  = lowbias32(key_site ^ (row * 0x85EBCA6B))
    h        = lowbias32(key_row ^ (col >> 1))        one hash per PAIR of columns (2k, 2k+1)
    r16      = h & 0xffff if col is even else h >> 16
    keep     = r16 >= round(rate * 2^16)              (clamped to [0, 65535])

Inverted dropout: kept elements are scaled by 1/(1-rate) (Keras semantics).

Attention-probability sites (the [T,T] score matrices) draw four decisions per hash (common.h rng_quad):

    h        = lowbias32(key_row ^ (col >> 2))        one hash per QUAD of keys (4k .. 4k+3)
    r8       = (h >> 8*(col & 3)) & 0xff
    keep     = r8 >= thr8 = round(rate * 2^8)          P(keep) = 1 - thr8/256
    scale    = 256 / (256 - thr8)                      = 1/P(keep): exact expectation for the quantised rate
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(16)
    return x


def site_key(seed, site):
    with np.errstate(over="ignore"):
        return lowbias32(np.uint32(seed & 0xFFFFFFFF) ^ (np.uint32(site) * np.uint32(0x9E3779B9)))


def threshold(rate):
    t = int(float(np.float32(rate)) * 65536.0 + 0.5)
    return np.uint32(min(max(t, 0), 65535))


def keep_mask(seed, site, rows, cols, rate):
    """Boolean keep-mask of shape [rows, cols] (row = 'hi' index, col = 'lo')."""
    ks = site_key(seed, site)
    with np.errstate(over="ignore"):
        r = np.arange(rows, dtype=np.uint32) * np.uint32(0x85EBCA6B)
    key_row = lowbias32(ks ^ r)                       # [rows]
    c = np.arange(cols, dtype=np.uint32)
    h = lowbias32(key_row[:, None] ^ (c[None, :] >> np.uint32(1)))      # [rows, cols]: one hash per column pair
    r16 = np.where((c[None, :] & np.uint32(1)) == 1, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return r16 >= threshold(rate)


def scaled_mask(seed, site, rows, cols, rate, dtype=np.float32):
    """Multiplicative inverted-dropout mask (0 or 1/(1-rate))."""
    if rate <= 0.0:
        return np.ones((rows, cols), dtype=dtype)
    k = keep_mask(seed, site, rows, cols, rate)
    return k.astype(dtype) * dtype(1.0 / (1.0 - rate))


def threshold8(rate):
    t = int(float(np.float32(rate)) * 256.0 + 0.5)
    return np.uint32(min(max(t, 0), 255))


def keep_mask_attn(seed, site, rows, cols, rate):
    """Keep-mask [rows, cols] of an attention-probability site: row = (b*H + h)*T + query, col = key (common.h rng_quad / rng_keep_q)."""
    ks = site_key(seed, site)
    with np.errstate(over="ignore"):
        r = np.arange(rows, dtype=np.uint32) * np.uint32(0x85EBCA6B)
    key_row = lowbias32(ks ^ r)
    c = np.arange(cols, dtype=np.uint32)
    h = lowbias32(key_row[:, None] ^ (c[None, :] >> np.uint32(2)))
    r8 = (h >> (np.uint32(8) * (c[None, :] & np.uint32(3)))) & np.uint32(0xFF)
    return r8 >= threshold8(rate)


def scaled_mask_attn(seed, site, rows, cols, rate, dtype=np.float32):
    """Multiplicative mask of an attention-probability site: 0 or 256/(256 - thr8)."""
    t8 = int(threshold8(rate))
    if rate <= 0.0 or t8 == 0:
        return np.ones((rows, cols), dtype=dtype)
    return keep_mask_attn(seed, site, rows, cols, rate).astype(dtype) * dtype(256.0 / (256.0 - t8))
