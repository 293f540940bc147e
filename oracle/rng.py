"""Counter-based RNG shared *by specification* between the HIP kernels and the oracle.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws dropout masks,
SpecAugment parameters and the teacher-forcing coin from TensorFlow's stateful RNG
(`tf.random.uniform`, Keras `Dropout`), whose bit stream cannot be reproduced; the
build replaces it with a stateless hash so that the HIP path and this oracle see
*identical* masks for a given (seed, stream, index).  Spec (all arithmetic mod 2^32):

    fmix32(x): x ^= x>>16; x *= 0x85EBCA6B; x ^= x>>13; x *= 0xC2B2AE35; x ^= x>>16
    k1 = fmix32(seed ^ (stream*0x9E3779B1 + 0x7F4A7C15))
    k2 = fmix32(k1 + 0x6A09E667 + stream)
    r(seed, stream, idx) = fmix32(((idx ^ k1) * 0x9E3779B1) + k2)

    dropout keep  : r >= floor(rate * 2^32)   (kept values are scaled by 1/(1-rate))
    uniform int   : (r * n) >> 32  in [0, n)
    uniform float : r * 2^-32      in [0, 1)
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _fmix32(x):
    x = np.asarray(x, dtype=np.uint64) & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M
    x ^= x >> np.uint64(16)
    return x


def keys(seed: int, stream: int):
    seed = np.uint64(seed & 0xFFFFFFFF)
    stream = np.uint64(stream & 0xFFFFFFFF)
    k1 = _fmix32(seed ^ ((stream * np.uint64(0x9E3779B1) + np.uint64(0x7F4A7C15)) & _M))
    k2 = _fmix32((k1 + np.uint64(0x6A09E667) + stream) & _M)
    return k1, k2


def rand_u32(seed: int, stream: int, idx):
    """r(seed, stream, idx) for an integer array `idx` -> uint64 array holding 32-bit values."""
    k1, k2 = keys(seed, stream)
    idx = np.asarray(idx, dtype=np.uint64) & _M
    return _fmix32((((idx ^ k1) * np.uint64(0x9E3779B1)) & _M) + k2)


def drop_threshold(rate: float) -> int:
    return int(np.floor(float(rate) * 4294967296.0)) & 0xFFFFFFFF if rate > 0 else 0


def dropout_mask(seed: int, stream: int, shape, rate: float, dtype=np.float64):
    """Inverted-dropout multiplier: 0 or 1/(1-rate), element index = C-order flat index."""
    n = int(np.prod(shape))
    if rate <= 0.0:
        return np.ones(shape, dtype=dtype)
    r = rand_u32(seed, stream, np.arange(n, dtype=np.uint64))
    keep = r >= np.uint64(drop_threshold(rate))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))  # kernels scale in f32
    return (keep.astype(dtype) * dtype(scale)).reshape(shape)


def uniform_int(seed: int, stream: int, idx: int, n: int) -> int:
    """Uniform integer in [0, n) (n >= 1); n <= 0 returns 0 (kernel does the same)."""
    if n <= 0:
        return 0
    r = int(rand_u32(seed, stream, np.array([idx]))[0])
    return (r * int(n)) >> 32


def uniform_float(seed: int, stream: int, idx: int) -> float:
    return float(int(rand_u32(seed, stream, np.array([idx]))[0])) * 2.0 ** -32
