"""TensorFlow tensor-bundle (V2 checkpoint) reader / writer without TensorFlow.

The reference checkpoints through ``tf.train.Saver()`` (edgegan/models/edgegan.py:421,635-657), i.e. files
``<dir>/EdgeGAN-Model-<step>.index`` + ``.data-00000-of-00001`` and a ``checkpoint`` state file.  This module
re-states the on-disk format so the engine can write checkpoints a TF-1.14 ``Saver.restore`` accepts and read
the published pretrained model (README.md:69):

* ``.index`` is a LevelDB-style sorted string table (tensorflow/core/lib/io/table_builder.cc, format.cc):
  prefix-compressed key/value blocks with restart arrays, a 5-byte trailer per block (compression type +
  masked CRC-32C), an index block, an (empty) metaindex block and a 48-byte footer ending in the magic
  0xdb4775248b80fb57.
* keys are variable names; values are ``BundleEntryProto`` messages (dtype, shape, shard, offset, size, masked
  CRC-32C of the bytes); key "" holds the ``BundleHeaderProto`` (tensorflow/core/protobuf/tensor_bundle.proto,
  tensorflow/core/util/tensor_bundle/tensor_bundle.cc).
* ``.data-XXXXX-of-YYYYY`` holds the raw little-endian tensor bytes at the recorded offsets.

Verified here by round trips, the format's own invariants (magic, per-block and per-tensor CRCs) and, byte for byte,
against a bundle laid out BY HAND from the format specification by an independent script
(tests/golden/make_bundle_golden.py -> tests/golden/bundle_golden.*).  It could not be cross-checked against
TensorFlow itself in this environment (no TF available) -- see DESIGN.md.
"""
from __future__ import annotations

import ctypes
import os
import re
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
BLOCK_SIZE = 262144          # table::Options::block_size default
RESTART_INTERVAL = 16

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BOOL = 1, 2, 3, 9, 10
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'),
           DT_INT64: np.dtype('<i8'), DT_BOOL: np.dtype('bool')}
_DTYPE_CODES = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
                np.dtype('int64'): DT_INT64, np.dtype('bool'): DT_BOOL}


# ------------------------------------------------------------------------------------------------
# CRC-32C (hardware path in libedgegan_hip.so; table fallback keeps this module importable anywhere)
# ------------------------------------------------------------------------------------------------
_crc_fn = None
_crc_table = None


def crc32c(data, crc: int = 0) -> int:
    global _crc_fn, _crc_table
    if _crc_fn is None:
        try:
            from ._lib import load_library
            lib = load_library()
            _crc_fn = lambda c, b, n: lib.eg_crc32c(c, b, n)
        except Exception:
            _crc_fn = False
    if isinstance(data, np.ndarray):
        buf = data if data.flags['C_CONTIGUOUS'] else data.copy(order='C')
        n = buf.nbytes
        if _crc_fn:
            return int(_crc_fn(crc, buf.ctypes.data_as(ctypes.c_void_p), n)) & 0xffffffff
        data = buf.tobytes()
    if _crc_fn:
        b = bytes(data)
        return int(_crc_fn(crc, ctypes.c_char_p(b), len(b))) & 0xffffffff
    if _crc_table is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            tab.append(c)
        _crc_table = tab
    c = crc ^ 0xffffffff
    for byte in bytes(data):
        c = _crc_table[(c ^ byte) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + MASK_DELTA) & 0xffffffff


def unmask_crc(m: int) -> int:
    rot = (m - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------
# varints / minimal protobuf
# ------------------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _pb_fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield field, wt, v


def _encode_shape(shape: Tuple[int, ...]) -> bytes:
    out = b''
    for d in shape:                                   # TensorShapeProto.dim = 2 ; Dim.size = 1
        dim = b'\x08' + _varint(int(d))
        out += b'\x12' + _varint(len(dim)) + dim
    return out


def _decode_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for f, _, v in _pb_fields(buf):
        if f == 2:
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def encode_entry(dtype: int, shape, shard_id: int, offset: int, size: int, crc_masked: int) -> bytes:
    out = b'\x08' + _varint(dtype)
    sh = _encode_shape(tuple(shape))
    out += b'\x12' + _varint(len(sh)) + sh
    if shard_id:
        out += b'\x18' + _varint(shard_id)
    if offset:
        out += b'\x20' + _varint(offset)
    if size:
        out += b'\x28' + _varint(size)
    out += b'\x35' + struct.pack('<I', crc_masked)     # field 6, fixed32
    return out


def decode_entry(buf: bytes) -> dict:
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            e['shape'] = _decode_shape(v)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['slices'] += 1
    return e


def encode_header(num_shards: int = 1) -> bytes:
    version = b'\x08\x01'                               # VersionDef.producer = 1 (kTensorBundleVersion)
    return b'\x08' + _varint(num_shards) + b'\x1a' + _varint(len(version)) + version


# ------------------------------------------------------------------------------------------------
# sorted string table
# ------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval: int):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.counter < self.interval:
            m = min(len(self.last_key), len(key))
            while shared < m and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self) -> bool:
        return not self.buf

    def finish(self) -> bytes:
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts)
        return out + struct.pack('<I', len(self.restarts))


def _write_block(f, contents: bytes) -> Tuple[int, int]:
    off = f.tell()
    trailer_type = b'\x00'                              # kNoCompression
    crc = mask_crc(crc32c(contents + trailer_type))
    f.write(contents + trailer_type + struct.pack('<I', crc))
    return off, len(contents)


def _shortest_separator(start: bytes, limit: bytes) -> bytes:
    """table_builder.cc FindShortestSeparator: a short key k with start <= k < limit (index key between two blocks)."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xff and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _short_successor(key: bytes) -> bytes:
    """table_builder.cc FindShortSuccessor: a short key >= key (index key of the last block)."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def write_table(path: str, items: List[Tuple[bytes, bytes]]):
    """TableBuilder::Add / Flush / Finish: the index entry of a flushed block is added when the NEXT key arrives (a
    shortest separator between the two) or at Finish (a short successor of the last key)."""
    items = sorted(items, key=lambda kv: kv[0])
    with open(path, 'wb') as f:
        index = _BlockBuilder(1)
        blk = _BlockBuilder(RESTART_INTERVAL)
        last_key = b''
        pending = None                                   # handle of a flushed block whose index entry is still due
        for k, v in items:
            if pending is not None:
                index.add(_shortest_separator(last_key, k), pending)
                pending = None
            blk.add(k, v)
            last_key = k
            if blk.size_estimate() >= BLOCK_SIZE:
                off, size = _write_block(f, blk.finish())
                pending = _varint(off) + _varint(size)
                blk = _BlockBuilder(RESTART_INTERVAL)
        if not blk.empty():
            off, size = _write_block(f, blk.finish())
            pending = _varint(off) + _varint(size)
        if pending is not None:
            index.add(_short_successor(last_key), pending)
        meta_off, meta_size = _write_block(f, _BlockBuilder(RESTART_INTERVAL).finish())
        idx_off, idx_size = _write_block(f, index.finish())
        footer = _varint(meta_off) + _varint(meta_size) + _varint(idx_off) + _varint(idx_size)
        footer += b'\x00' * (40 - len(footer))
        f.write(footer + struct.pack('<Q', TABLE_MAGIC))


def _read_block(buf: bytes, off: int, size: int, verify: bool = True) -> bytes:
    contents = buf[off:off + size]
    ctype = buf[off + size]
    if verify:
        want = struct.unpack_from('<I', buf, off + size + 1)[0]
        if unmask_crc(want) != crc32c(contents + bytes([ctype])):
            raise ValueError('table block checksum mismatch')
    if ctype != 0:
        raise ValueError(f'compressed table block (type {ctype}) is not supported')
    return contents


def _iter_block(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f'{path}: not an sstable (bad magic)')
    footer = buf[-48:]
    _, p = _read_varint(footer, 0)
    _, p = _read_varint(footer, p)
    idx_off, p = _read_varint(footer, p)
    idx_size, p = _read_varint(footer, p)
    out = []
    for _, handle in _iter_block(_read_block(buf, idx_off, idx_size, verify)):
        off, q = _read_varint(handle, 0)
        size, _ = _read_varint(handle, q)
        out.extend(_iter_block(_read_block(buf, off, size, verify)))
    return out


# ------------------------------------------------------------------------------------------------
# bundle
# ------------------------------------------------------------------------------------------------
def data_filename(prefix: str, shard: int = 0, num_shards: int = 1) -> str:
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]):
    """Writes <prefix>.index and <prefix>.data-00000-of-00001 (one shard, like a single-device Saver)."""
    items = [(b'', encode_header(1))]
    offset = 0
    tmp = data_filename(prefix) + '.tempstate'
    with open(tmp, 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            if not a.flags['C_CONTIGUOUS']:
                a = a.copy(order='C')
            code = _DTYPE_CODES.get(a.dtype)
            if code is None:
                raise ValueError(f'{name}: unsupported dtype {a.dtype}')
            a = a.astype(_DTYPES[code], copy=False)
            f.write(a.tobytes() if a.ndim == 0 else memoryview(a.reshape(-1)).cast('B'))
            items.append((name.encode(), encode_entry(code, a.shape, 0, offset, a.nbytes, mask_crc(crc32c(a)))))
            offset += a.nbytes
    os.replace(tmp, data_filename(prefix))
    write_table(prefix + '.index', items)


def read_bundle(prefix: str, verify: bool = True, names: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    entries = read_table(prefix + '.index', verify)
    num_shards = 1
    out: Dict[str, np.ndarray] = {}
    want = set(names) if names is not None else None
    files = {}
    for k, v in entries:
        if k == b'':
            for f, _, val in _pb_fields(v):
                if f == 1:
                    num_shards = val
                elif f == 2 and val != 0:
                    raise ValueError('big-endian bundles are not supported')
            continue
        name = k.decode()
        if want is not None and name not in want:
            continue
        e = decode_entry(v)
        if e['slices']:
            raise ValueError(f'{name}: partitioned (sliced) variables are not supported')
        if e['dtype'] not in _DTYPES:
            continue                                   # e.g. string tensors of a Saver's bookkeeping
        path = data_filename(prefix, e['shard_id'], num_shards)
        if path not in files:
            files[path] = np.memmap(path, dtype=np.uint8, mode='r')
        raw = np.asarray(files[path][e['offset']:e['offset'] + e['size']])
        if verify and unmask_crc(e['crc32c']) != crc32c(raw):
            raise ValueError(f'{name}: tensor checksum mismatch')
        out[name] = raw.view(_DTYPES[e['dtype']]).reshape(e['shape']).copy()
    return out


def list_bundle(prefix: str) -> Dict[str, Tuple[Tuple[int, ...], int]]:
    return {k.decode(): (decode_entry(v)['shape'], decode_entry(v)['dtype']) for k, v in read_table(prefix + '.index')
            if k != b''}


# ------------------------------------------------------------------------------------------------
# `checkpoint` state file (CheckpointState text proto, tf.train.get_checkpoint_state)
# ------------------------------------------------------------------------------------------------
def write_checkpoint_state(directory: str, latest: str, all_paths: List[str]):
    with open(os.path.join(directory, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % latest)
        for p in all_paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)


def read_checkpoint_state(directory: str) -> Tuple[Optional[str], List[str]]:
    path = os.path.join(directory, 'checkpoint')
    if not os.path.exists(path):
        return None, []
    latest, allp = None, []
    for line in open(path):
        m = re.match(r'\s*(\w+)\s*:\s*"(.*)"\s*$', line)
        if not m:
            continue
        if m.group(1) == 'model_checkpoint_path':
            latest = m.group(2)
        elif m.group(1) == 'all_model_checkpoint_paths':
            allp.append(m.group(2))
    return latest, allp
