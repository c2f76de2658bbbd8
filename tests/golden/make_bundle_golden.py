#!/usr/bin/env python
"""A TensorFlow tensor-bundle (V2 checkpoint) written out BY HAND, byte by byte, from the format's specification --
independently of edgegan_amd/tf_bundle.py (nothing is imported from the package; the CRC, the varints, the protobuf
wire encoding and the table layout are all restated here in straight-line code).  tests/test_checkpoint_cpu.py checks
that tf_bundle.write_bundle produces exactly these bytes and that tf_bundle.read_bundle reads them.

Content: two float32 variables,  "a" = [1.0, 2.0]  (shape [2])  and  "b/c" = [[3.0, -4.5]]  (shape [1, 2]).

Spec followed (TensorFlow 1.14 sources):
  tensorflow/core/util/tensor_bundle/tensor_bundle.cc  BundleWriter: header entry under key "", entries sorted by key,
      tensor bytes appended to <prefix>.data-00000-of-00001, per-tensor masked crc32c in the entry;
  tensorflow/core/protobuf/tensor_bundle.proto         BundleHeaderProto{num_shards=1, endianness=2, version=3},
      BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)};
  tensorflow/core/lib/io/table_builder.cc, block_builder.cc, format.cc (LevelDB table): data block = entries
      (shared, non_shared, value_len varints + key suffix + value) + restart array + restart count; every block is
      followed by a 1-byte compression type (0) and the masked crc32c of block+type; then an empty metaindex block,
      the index block (key = short successor of the last key, value = BlockHandle varints), and the 48-byte footer
      (metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 little-endian);
  tensorflow/core/lib/hash/crc32c.h                    Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8.
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bit by bit."""
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def varint(n: int) -> bytes:
    out = b''
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


assert crc32c(b'123456789') == 0xE3069283            # the standard CRC-32C check value (RFC 3720 appendix)

# ---- the two tensors and the data file -------------------------------------------------------------------------
a_bytes = struct.pack('<2f', 1.0, 2.0)              # "a",   shape [2]
bc_bytes = struct.pack('<2f', 3.0, -4.5)            # "b/c", shape [1, 2]
data_file = a_bytes + bc_bytes                       # offsets 0 and 8

# ---- protobuf values -------------------------------------------------------------------------------------------
# BundleHeaderProto: num_shards (field 1, varint) = 1; endianness (field 2) = LITTLE = 0 -> default, not serialised;
#                    version (field 3, message VersionDef{producer (field 1) = 1})
header = b'\x08\x01' + b'\x1a\x02' + b'\x08\x01'


def entry(dims, offset, size, payload):
    # dtype (field 1) = DT_FLOAT = 1
    e = b'\x08\x01'
    # shape (field 2, TensorShapeProto): repeated dim (field 2, Dim{size (field 1)})
    shape = b''.join(b'\x12' + varint(len(d)) + d for d in (b'\x08' + varint(n) for n in dims))
    e += b'\x12' + varint(len(shape)) + shape
    # shard_id (field 3) = 0 -> default, not serialised; offset (field 4) only when non-zero
    if offset:
        e += b'\x20' + varint(offset)
    e += b'\x28' + varint(size)                      # size (field 5)
    e += b'\x35' + struct.pack('<I', masked(crc32c(payload)))   # crc32c (field 6, fixed32)
    return e


entries = [(b'', header), (b'a', entry([2], 0, 8, a_bytes)), (b'b/c', entry([1, 2], 8, 8, bc_bytes))]


# ---- the table (.index) ----------------------------------------------------------------------------------------
def block(kvs):
    """One block with restart interval 16: < 16 entries -> a single restart point at offset 0."""
    out, last = b'', b''
    for k, v in kvs:
        shared = 0
        if out:                                      # prefix compression against the previous key
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    return out + struct.pack('<I', 0) + struct.pack('<I', 1)       # restart[0] = 0, num_restarts = 1


def with_trailer(b):
    return b + b'\x00' + struct.pack('<I', masked(crc32c(b + b'\x00')))     # type 0 = no compression


data_block = block(entries)
meta_block = block([])                               # empty metaindex block
index_file = with_trailer(data_block)
data_handle = varint(0) + varint(len(data_block))
meta_off = len(index_file)
index_file += with_trailer(meta_block)
meta_handle = varint(meta_off) + varint(len(meta_block))
# index block: one entry; key = FindShortSuccessor("b/c") = "c" (first byte that can be incremented, then truncate)
index_block = block([(b'c', data_handle)])
index_off = len(index_file)
index_file += with_trailer(index_block)
index_handle = varint(index_off) + varint(len(index_block))
footer = meta_handle + index_handle
footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
index_file += footer

if __name__ == '__main__':
    with open(os.path.join(HERE, 'bundle_golden.index'), 'wb') as f:
        f.write(index_file)
    with open(os.path.join(HERE, 'bundle_golden.data-00000-of-00001'), 'wb') as f:
        f.write(data_file)
    print('index', len(index_file), 'bytes:', index_file.hex())
    print('data ', len(data_file), 'bytes:', data_file.hex())
