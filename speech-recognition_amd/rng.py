"""RNG stream ids of the stateless dropout / SpecAugment / teacher-forcing draws.

The reference draws from TensorFlow's stateful RNG (Keras `Dropout`, `tf.random.uniform`), which
cannot be reproduced bit for bit; the build uses a counter-based hash r(seed, stream, index)
(device code: csrc/common.h; specification and numpy mirror: oracle/rng.py) so that masks are a pure
function of (seed, site, element) - reproducible, graph-replayable, and recomputed in the backward
pass instead of being stored.
"""
STREAM_CONV1_DROP = 1      # Listener.dropout after conv1 (las.py:183), flat index over [B,T1,F1,32]
STREAM_CONV2_DROP = 2      # Listener.dropout after conv2 (las.py:184)
STREAM_SPECAUG = 3         # data.py:282-301 draws, index = clip*64 + draw
STREAM_TEACHER = 4         # las.py:366 one coin per batch
STREAM_ENC_IN = 10         # + 2*layer + direction : Keras RNN input dropout, index over [B, Din]
STREAM_ENC_REC = 60        # + 2*layer + direction : recurrent dropout, index over [B, H]
# recurrent_dropout != 0 puts the Keras cells into implementation 1 - one mask PER GATE on both operands: gate g of a layer draws from
# STREAM_ENC_IN / STREAM_ENC_REC + 2*layer + direction + 128*g (g = 0 is the single-mask stream)
STREAM_DEC = 1000          # + 32*step + {0: embedding dropout, 1: output dropout, 2+j: decoder layer j input dropout}
DEC_STREAMS_PER_STEP = 32
MAX_DECODER_LAYERS = DEC_STREAMS_PER_STEP - 2


def fmix32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def rand_u32(seed: int, stream: int, idx: int) -> int:
    """Host mirror of the device hash (used for the teacher-forcing coin, drawn on the host)."""
    k1 = fmix32((seed & 0xFFFFFFFF) ^ ((stream * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF))
    k2 = fmix32((k1 + 0x6A09E667 + stream) & 0xFFFFFFFF)
    return fmix32(((((idx ^ k1) & 0xFFFFFFFF) * 0x9E3779B1) & 0xFFFFFFFF) + k2)


def uniform_float(seed: int, stream: int, idx: int) -> float:
    return rand_u32(seed, stream, idx) * 2.0 ** -32
