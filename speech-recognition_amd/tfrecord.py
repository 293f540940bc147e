"""GZIP TFRecord files of (feature tensor, token tensor) examples, without TensorFlow.

Format (written by the reference's run/make_tfrecord.py:39-58, read by data.py:64-79): a gzip stream of
records ``u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)``; each payload
is a serialized string TensorProto with two elements, the serialized float32 feature TensorProto
([T, F, 1]) and the serialized int32 token TensorProto ([U]).  Only the TensorProto fields those two
writers emit are handled (dtype, tensor_shape, tensor_content, string_val and the packed
float_val / int_val forms small tensors may take).

Records are streamed: a file is never held in memory as a whole.
"""
import ctypes as C
import gzip
import struct
from typing import Iterator, Tuple

import numpy as np

DT_FLOAT, DT_INT32, DT_STRING = 1, 3, 7
_NP = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4")}


# ------------------------------------------------------------------------------------------ checksums
def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli); runs in the native library (asr_crc32c)."""
    from ._lib import load
    data = bytes(data)
    return int(load().asr_crc32c(data, len(data), crc)) & 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ protobuf wire
def _read_varint(buf, pos):
    result = shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if byte < 0x80:
            return result, pos
        shift += 7


def _write_varint(value: int) -> bytes:
    out = bytearray()
    while True:
        low = value & 0x7F
        value >>= 7
        if value:
            out.append(low | 0x80)
        else:
            out.append(low)
            return bytes(out)


def _walk(buf):
    """(field number, wire type, value) of one message level; value is an int or a memoryview."""
    view, pos, end = memoryview(buf), 0, len(buf)
    while pos < end:
        key, pos = _read_varint(view, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _read_varint(view, pos)
        elif wire == 2:
            size, pos = _read_varint(view, pos)
            value, pos = view[pos:pos + size], pos + size
        elif wire == 5:
            value, pos = view[pos:pos + 4], pos + 4
        elif wire == 1:
            value, pos = view[pos:pos + 8], pos + 8
        else:
            raise ValueError(f"TensorProto: unsupported wire type {wire}")
        yield field, wire, value


def _len_field(field: int, payload: bytes) -> bytes:
    return _write_varint((field << 3) | 2) + _write_varint(len(payload)) + payload


def parse_tensor(buf):
    """Serialized TensorProto -> ndarray (float32 / int32) or list of bytes (string tensor)."""
    dtype, dims, content, strings, scalars = None, [], None, [], []
    for field, wire, value in _walk(buf):
        if field == 1:
            dtype = value
        elif field == 2:                                   # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, dim in _walk(value):
                if f2 == 2:
                    dims.append(next((v for f3, _, v in _walk(dim) if f3 == 1), 0))
        elif field == 4:
            content = bytes(value)
        elif field == 8:
            strings.append(bytes(value))
        elif field == 5:                                   # float_val (packed or single fixed32)
            scalars.extend(np.frombuffer(bytes(value), "<f4").tolist())
        elif field == 7:                                   # int_val (packed varints or a single varint)
            if wire == 2:
                pos = 0
                while pos < len(value):
                    v, pos = _read_varint(value, pos)
                    scalars.append(v - (1 << 64) if v >> 63 else v)
            else:
                scalars.append(value - (1 << 64) if value >> 63 else value)
    if dtype == DT_STRING:
        return strings
    if dtype not in _NP:
        raise ValueError(f"TensorProto: unsupported dtype enum {dtype}")
    count = int(np.prod(dims)) if dims else 1
    if content is not None:
        arr = np.frombuffer(content, _NP[dtype])
    else:
        arr = np.asarray(scalars, _NP[dtype])
        if arr.size == 1 and count > 1:                    # TensorProto's "repeat the last value" compression
            arr = np.full(count, arr[0], _NP[dtype])
    return arr.reshape(dims)


def serialize_tensor(value) -> bytes:
    """ndarray (float32 / int32) or list of bytes -> serialized TensorProto (tf.io.serialize_tensor layout)."""
    if isinstance(value, (list, tuple)) and all(isinstance(v, (bytes, bytearray)) for v in value):
        shape = _len_field(2, _len_field(2, _write_varint((1 << 3) | 0) + _write_varint(len(value))))
        return _write_varint((1 << 3) | 0) + _write_varint(DT_STRING) + shape + b"".join(_len_field(8, bytes(v)) for v in value)
    arr = np.asarray(value)
    if arr.dtype == np.float32:
        dtype = DT_FLOAT
    elif arr.dtype == np.int32:
        dtype = DT_INT32
    else:
        raise ValueError(f"serialize_tensor: unsupported dtype {arr.dtype}")
    dims = b"".join(_len_field(2, _write_varint((1 << 3) | 0) + _write_varint(int(d))) for d in arr.shape)
    out = _write_varint((1 << 3) | 0) + _write_varint(dtype) + _len_field(2, dims)
    return out + _len_field(4, np.ascontiguousarray(arr).astype(_NP[dtype], copy=False).tobytes())


# ------------------------------------------------------------------------------------------ record framing
def read_records(path: str, check_crc: bool = False) -> Iterator[bytes]:
    with gzip.open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated TFRecord header")
            (length,), (len_crc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            payload = f.read(length)
            tail = f.read(4)
            if len(payload) < length or len(tail) < 4:
                raise ValueError(f"{path}: truncated TFRecord payload")
            if check_crc:
                if masked_crc(head[:8]) != len_crc or masked_crc(payload) != struct.unpack("<I", tail)[0]:
                    raise ValueError(f"{path}: TFRecord checksum mismatch")
            yield payload


def read_examples(path: str, check_crc: bool = False) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
    """Yield (features f32 [T, F, 1], tokens i32 [U]) per record (data.py:68-78)."""
    for payload in read_records(path, check_crc):
        parts = parse_tensor(payload)
        yield parse_tensor(parts[0]), parse_tensor(parts[1])


class TFRecordWriter:
    """GZIP TFRecord writer (run/make_tfrecord.py:47-58): ``write(features, tokens)`` per example."""

    def __init__(self, path: str):
        self._f = gzip.open(path, "wb")

    def write_record(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload)))

    def write(self, features: np.ndarray, tokens: np.ndarray):
        self.write_record(serialize_tensor([serialize_tensor(np.asarray(features, np.float32)),
                                            serialize_tensor(np.asarray(tokens, np.int32))]))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
