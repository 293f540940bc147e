"""Oracle (TEST INFRASTRUCTURE): reader for the reference's GZIP TFRecord fixture
(tests/data/wav_dataset.tfrecord, written by run/make_tfrecord.py:39-58 and read back by
data.py:64-79: each record is a serialized string TensorProto holding two serialized
TensorProtos, the float32 feature tensor and the int32 token tensor).  No TensorFlow needed.
"""
import gzip
import struct

import numpy as np


def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, i
        s += 7


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            n, i = _varint(buf, i)
            v = buf[i:i + n]
            i += n
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        else:
            raise ValueError(f"wire type {wt}")
        yield fno, wt, v


_DT = {1: np.float32, 3: np.int32, 9: np.int64, 7: None}


def parse_tensor_proto(buf):
    dtype, shape, content, strings, floats, ints = None, [], None, [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3
                    shape.append(size)
        elif fno == 4:
            content = bytes(v)
        elif fno == 8:
            strings.append(bytes(v))
        elif fno == 5:
            floats.extend(np.frombuffer(bytes(v), np.float32).tolist() if wt == 2 else [struct.unpack("<f", v)[0]])
        elif fno == 7:
            if wt == 2:
                j = 0
                while j < len(v):
                    x, j = _varint(v, j)
                    ints.append(x)
            else:
                ints.append(v)
    if dtype == 7:
        return strings
    np_dt = _DT[dtype]
    if content is not None:
        return np.frombuffer(content, np_dt).reshape(shape)
    vals = floats if np_dt == np.float32 else ints
    arr = np.array(vals, np_dt)
    n = int(np.prod(shape)) if shape else 1
    if arr.size == 1 and n > 1:
        arr = np.full(n, arr[0], np_dt)
    return arr.reshape(shape)


def read_tfrecord(path):
    """Yield (features float32 [T, F, 1], tokens int32 [U]) per record."""
    data = gzip.open(path, "rb").read()
    i = 0
    while i < len(data):
        (n,) = struct.unpack("<Q", data[i:i + 8])
        rec = data[i + 12:i + 12 + n]
        i += 12 + n + 4
        parts = parse_tensor_proto(rec)
        yield parse_tensor_proto(parts[0]), parse_tensor_proto(parts[1])
