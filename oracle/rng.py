"""Counter-based RNG shared (bit-exactly) by the oracle and the HIP library.

The reference seeds nothing, so the build defines its own seeded,
order-independent generator.  All three random decisions on the hot path draw
from it:

* weight init   (Keras glorot_uniform, nsga_penalty.py:255 defaults)   stream 0x1000 + tensor index
* dropout masks (Dropout(0.3), nsga_penalty.py:323)                     stream 0x2000 + fc-layer index
* epoch shuffle (Model.fit shuffle=True, nsga_penalty.py:383)           stream 0x3000

``rng_u32(seed, stream, ctr, idx)`` is four chained murmur3 finalisers; the C++
twin is ``cmoop_rng_u32`` in cmoop_audio_processing_amd/csrc/rng.h.
"""
import numpy as np

STREAM_INIT = 0x1000
STREAM_DROPOUT = 0x2000
STREAM_SHUFFLE = 0x3000

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def rng_u32(seed, stream, ctr, idx):
    """uint32 hash of (seed, stream, ctr, idx); idx may be an array."""
    h = _fmix32((np.uint64(int(seed) & 0xFFFFFFFF) + np.uint64(0x9E3779B9)) & _M32)
    h = _fmix32(h ^ np.uint64(int(stream) & 0xFFFFFFFF))
    h = _fmix32(h ^ np.uint64(int(ctr) & 0xFFFFFFFF))
    h = _fmix32(h ^ (np.asarray(idx, dtype=np.uint64) & _M32))
    return h.astype(np.uint32)


def glorot_uniform(seed, tensor_index, shape, fan_in, fan_out):
    """Keras glorot_uniform: U(-limit, limit), limit = sqrt(6/(fan_in+fan_out)).

    Element i (row-major over the CANONICAL layout) is
    ``float32(2*u24 - 2**24) * float32(limit / 2**24)`` -- one fp32 rounding, so
    host C++, device and numpy agree bit for bit.
    """
    n = int(np.prod(shape))
    u24 = (rng_u32(seed, STREAM_INIT + tensor_index, 0, np.arange(n, dtype=np.uint64)) >> np.uint32(8)).astype(np.int64)
    limit = np.sqrt(6.0 / float(fan_in + fan_out))
    scale = np.float32(limit / 16777216.0)
    s = (2 * u24 - 16777216).astype(np.float32)
    return (s * scale).astype(np.float32).reshape(shape)


def dropout_threshold(rate):
    return int(float(rate) * 16777216.0)


def dropout_keep(seed, layer_index, step, n_rows, n_cols, rate):
    """bool [n_rows, n_cols]; keep iff u24 >= floor(rate * 2**24)."""
    idx = np.arange(n_rows * n_cols, dtype=np.uint64)
    u24 = rng_u32(seed, STREAM_DROPOUT + layer_index, step, idx) >> np.uint32(8)
    return (u24 >= np.uint32(dropout_threshold(rate))).reshape(n_rows, n_cols)


def epoch_permutation(seed, epoch, n):
    """Permutation of range(n): argsort of (hash key, index) -- unique keys."""
    i = np.arange(n, dtype=np.uint64)
    key = (rng_u32(seed, STREAM_SHUFFLE, epoch, i).astype(np.uint64) << np.uint64(32)) | i
    return np.argsort(key, kind="stable").astype(np.int32)
