"""TensorFlow tensor-bundle checkpoints (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``) without
TensorFlow: the format `model.save_weights(path)` / `model.load_weights(path)` use in run/train.py:152-154,
208-212, so that checkpoints written by the reference load into this build's models and vice versa.

``.index`` is a LevelDB-style sorted string table: data blocks of prefix-compressed (key, value) entries
with a restart array, each followed by a 5-byte trailer (compression type, masked CRC-32C), an index block
pointing at the data blocks, and a 48-byte footer (metaindex handle, index handle, magic).  The entry
with the empty key holds BundleHeaderProto, every other key is a variable name whose value is a
BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.  Tensor bytes live in the
data shard at [offset, offset+size), little-endian, row-major.
"""
import os
import struct
from typing import Dict, Iterator, Tuple

import numpy as np

from .tfrecord import _len_field, _read_varint, _walk, _write_varint, crc32c

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 10: np.dtype("bool"), 19: np.dtype("<f2")}
_DTYPE_ENUM = {v: k for k, v in _DTYPES.items()}


def _mask(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _unmask(m: int) -> int:
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ table reading
def _block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    body, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(body) < size or len(trailer) < 5:
        raise ValueError("checkpoint index: block runs past the end of the file")
    if trailer[0] != 0:
        raise NotImplementedError("checkpoint index: compressed table blocks are not supported (TF writes them uncompressed)")
    if verify and _unmask(struct.unpack("<I", trailer[1:])[0]) != crc32c(body + trailer[:1]):
        raise ValueError("checkpoint index: block checksum mismatch")
    return body


def _entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    (num_restarts,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    view = memoryview(block)
    while pos < end:
        shared, pos = _read_varint(view, pos)
        non_shared, pos = _read_varint(view, pos)
        value_len, pos = _read_varint(view, pos)
        key = key[:shared] + bytes(view[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(view[pos:pos + value_len])
        pos += value_len


def _handle(buf, pos):
    offset, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return offset, size, pos


def read_index(index_path: str, verify: bool = True) -> Dict[str, dict]:
    """name -> dict(dtype, shape, shard_id, offset, size, crc32c); the '' key holds the header fields."""
    with open(index_path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise ValueError(f"{index_path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = memoryview(buf)[-48:]
    _, _, pos = _handle(footer, 0)                       # metaindex handle (unused)
    idx_off, idx_size, _ = _handle(footer, pos)
    out = {}
    for _, handle in _entries(_block(buf, idx_off, idx_size, verify)):
        off, size, _ = _handle(memoryview(handle), 0)
        for key, value in _entries(_block(buf, off, size, verify)):
            name = key.decode()
            if name == "":
                out[""] = {f: v for f, _, v in _walk(value) if isinstance(v, int)}
                continue
            e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None)
            for field, wire, v in _walk(value):
                if field == 1:
                    e["dtype"] = v
                elif field == 2:
                    for f2, _, dim in _walk(v):
                        if f2 == 2:
                            e["shape"].append(next((x for f3, _, x in _walk(dim) if f3 == 1), 0))
                elif field == 3:
                    e["shard_id"] = v
                elif field == 4:
                    e["offset"] = v
                elif field == 5:
                    e["size"] = v
                elif field == 6:
                    e["crc32c"] = struct.unpack("<I", bytes(v))[0]
            out[name] = e
    return out


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """All numeric tensors of the checkpoint `prefix` (string-typed bookkeeping entries are skipped)."""
    index = read_index(prefix + ".index", verify)
    shards = index.get("", {}).get(1, 1) or 1
    out, files = {}, {}
    try:
        for name, e in index.items():
            if name == "" or e["dtype"] not in _DTYPES:
                continue
            path = f"{prefix}.data-{e['shard_id']:05d}-of-{shards:05d}"
            if path not in files:
                files[path] = open(path, "rb")
            f = files[path]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError(f"{path}: tensor {name!r} runs past the end of the shard")
            if verify and e["crc32c"] is not None and _unmask(e["crc32c"]) != crc32c(raw):
                raise ValueError(f"{path}: checksum mismatch in tensor {name!r}")
            out[name] = np.frombuffer(raw, _DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ------------------------------------------------------------------------------------------ object graph
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def _read_string_scalar(prefix: str, entry: dict, shards: int, verify: bool) -> bytes:
    """A scalar DT_STRING tensor in the data shard: varint length, masked CRC-32C of the length as a
    uint32, the bytes; the entry checksum covers uint32 length + those 4 CRC bytes + the bytes."""
    with open(f"{prefix}.data-{entry['shard_id']:05d}-of-{shards:05d}", "rb") as f:
        f.seek(entry["offset"])
        raw = f.read(entry["size"])
    n, pos = _read_varint(raw, 0)
    body = raw[pos + 4:pos + 4 + n]
    if verify:
        packed = struct.pack("<I", n)
        if _unmask(struct.unpack("<I", raw[pos:pos + 4])[0]) != crc32c(packed):
            raise ValueError("checkpoint: string tensor length checksum mismatch")
        if entry["crc32c"] is not None and _unmask(entry["crc32c"]) != crc32c(packed + raw[pos:pos + 4] + body):
            raise ValueError("checkpoint: string tensor checksum mismatch")
    return body


def read_object_graph(prefix: str, verify: bool = True):
    """TrackableObjectGraph of a checkpoint: list of nodes, each dict(children={local_name: node_id},
    attributes=[dict(name, full_name, checkpoint_key)]).  None when the checkpoint has no object graph."""
    index = read_index(prefix + ".index", verify)
    if OBJECT_GRAPH_KEY not in index:
        return None
    blob = _read_string_scalar(prefix, index[OBJECT_GRAPH_KEY], index.get("", {}).get(1, 1) or 1, verify)
    nodes = []
    for field, _, node in _walk(blob):
        if field != 1:
            continue
        children, attributes = {}, []
        for f2, _, v in _walk(node):
            if f2 == 1:                                   # ObjectReference {node_id = 1, local_name = 2}
                ref = {f3: x for f3, _, x in _walk(v)}
                children[bytes(ref.get(2, b"")).decode()] = ref.get(1, 0)
            elif f2 == 2:                                 # SerializedTensor {name = 1, full_name = 2, checkpoint_key = 3}
                t = {f3: bytes(x).decode() for f3, _, x in _walk(v)}
                attributes.append(dict(name=t.get(1, ""), full_name=t.get(2, ""), checkpoint_key=t.get(3, "")))
        nodes.append(dict(children=children, attributes=attributes))
    return nodes


def build_object_graph(variable_names) -> bytes:
    """TrackableObjectGraph for variables named by their attribute path from the model root
    ('listener/conv1/kernel'): one node per path component, the leaf carrying the VARIABLE_VALUE
    attribute - the dependency tree Keras' object-based `load_weights` walks by local name."""
    nodes = [dict(children={}, attr=None)]
    for name in sorted(variable_names):
        cur = 0
        for part in name.split("/"):
            nxt = nodes[cur]["children"].get(part)
            if nxt is None:
                nxt = len(nodes)
                nodes.append(dict(children={}, attr=None))
                nodes[cur]["children"][part] = nxt
            cur = nxt
        nodes[cur]["attr"] = name
    out = b""
    for node in nodes:
        body = b""
        for local_name, node_id in node["children"].items():
            ref = (_write_varint(8) + _write_varint(node_id) if node_id else b"") + _len_field(2, local_name.encode())
            body += _len_field(1, ref)
        if node["attr"] is not None:
            t = _len_field(1, b"VARIABLE_VALUE") + _len_field(2, node["attr"].encode()) + _len_field(3, (node["attr"] + _SUFFIX).encode())
            body += _len_field(2, t)
        out += _len_field(1, body)
    return out


def load_variables(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """Variables of a Keras `save_weights` checkpoint keyed by attribute path ('listener/conv1/kernel')."""
    return {k[:-len(_SUFFIX)]: v for k, v in read_bundle(prefix, verify).items() if k.endswith(_SUFFIX)}


def save_variables(prefix: str, variables: Dict[str, np.ndarray]):
    """Write variables keyed by attribute path the way Keras `save_weights(prefix)` lays them out
    (checkpoint keys '<path>/.ATTRIBUTES/VARIABLE_VALUE' plus the object graph)."""
    write_bundle(prefix, {k + _SUFFIX: np.asarray(v) for k, v in variables.items()}, build_object_graph(variables.keys()))


# ------------------------------------------------------------------------------------------ table writing
def _put_block(out: bytearray, entries) -> Tuple[int, int]:
    """One block with a restart point at every entry (no prefix sharing): valid, if not the most compact."""
    body, restarts = bytearray(), []
    for key, value in entries:
        restarts.append(len(body))
        body += _write_varint(0) + _write_varint(len(key)) + _write_varint(len(value)) + key + value
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", max(len(restarts), 1))
    offset = len(out)
    out += body + b"\0" + struct.pack("<I", _mask(crc32c(bytes(body) + b"\0")))
    return offset, len(body)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], object_graph: bytes = None):
    """Write `tensors` (and, if given, the serialized TrackableObjectGraph) as a one-shard tensor bundle in
    the layout tf.train.load_checkpoint / Keras load_weights read."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    entries = []
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        offset = 0
        if object_graph is not None:                      # '_' sorts before every lower-case variable path
            packed = struct.pack("<I", len(object_graph))
            len_crc = struct.pack("<I", _mask(crc32c(packed)))
            raw = _write_varint(len(object_graph)) + len_crc + object_graph
            data.write(raw)
            value = _write_varint(8) + _write_varint(7) + _len_field(2, b"") + _write_varint(5 << 3) + _write_varint(len(raw))
            value += _write_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc32c(packed + len_crc + object_graph)))
            entries.append((OBJECT_GRAPH_KEY.encode(), value))
            offset += len(raw)
        for name in sorted(tensors):
            arr = np.ascontiguousarray(tensors[name])
            if arr.dtype not in _DTYPE_ENUM:
                raise ValueError(f"write_bundle: unsupported dtype {arr.dtype} for {name!r}")
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            shape = b"".join(_len_field(2, _write_varint(8) + _write_varint(int(d))) for d in arr.shape)
            value = _write_varint(8) + _write_varint(_DTYPE_ENUM[arr.dtype]) + _len_field(2, shape)
            if offset:
                value += _write_varint(4 << 3) + _write_varint(offset)
            value += _write_varint(5 << 3) + _write_varint(len(raw))
            value += _write_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc32c(raw)))
            entries.append((name.encode(), value))
            offset += len(raw)
    entries.sort(key=lambda kv: kv[0])                    # table keys are in bytewise order
    header = _write_varint(8) + _write_varint(1) + _len_field(3, _write_varint(8) + _write_varint(1))   # num_shards=1, version{producer=1}
    out = bytearray()
    d_off, d_size = _put_block(out, [(b"", header)] + entries)
    m_off, m_size = _put_block(out, [])
    last_key = entries[-1][0] if entries else b""
    i_off, i_size = _put_block(out, [(last_key + b"\xff", _write_varint(d_off) + _write_varint(d_size))])
    footer = _write_varint(m_off) + _write_varint(m_size) + _write_varint(i_off) + _write_varint(i_size)
    out += footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(out)
