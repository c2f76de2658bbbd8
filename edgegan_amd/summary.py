"""TensorBoard event files for the scalar summaries of the reference (edgegan/models/edgegan.py:351-365,386-407),
without TensorFlow.  The reference spends two extra forward passes per iteration on ``sess.run(d_sum / g_sum)``
(edgegan.py:427-433); here the scalars come out of the training step itself (``eg_read_losses``).

File format (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/util/event.proto): a sequence of records
``uint64 length | masked crc32c(length) | data | masked crc32c(data)`` whose data are ``Event`` protos;
the first event carries ``file_version = "brain.Event:2"``.  Histogram and image summaries are not written."""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, Iterable, List, Tuple

from .tf_bundle import crc32c, mask_crc, unmask_crc, _varint, _pb_fields

# tags exactly as the reference names them (d_sum and g_sum members)
SCALAR_TAGS = ['joint_dis_dloss', 'loss_d_ac', 'image_dis_dloss', 'edge_dis_dloss',          # d_sum
               'zl_loss', 'joint_dis_gloss', 'loss_g_ac', 'edge_gloss', 'image_gloss',       # g_sum
               'image_dis_gloss', 'edge_dis_gloss']


def _record(data: bytes) -> bytes:
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data)))


def _event(wall_time: float, step: int, payload: bytes) -> bytes:
    out = b'\x09' + struct.pack('<d', wall_time)           # wall_time = 1 (double)
    if step:
        out += b'\x10' + _varint(step)                     # step = 2 (int64)
    return out + payload


class SummaryWriter:
    """tf.summary.FileWriter(logdir) stand-in (edgegan.py:443)."""

    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        ver = b'brain.Event:2'
        self._f.write(_record(_event(time.time(), 0, b'\x1a' + _varint(len(ver)) + ver)))   # file_version = 3
        self._f.flush()

    def add_scalars(self, values: Dict[str, float], step: int):
        summary = b''
        for tag, v in values.items():
            t = tag.encode()
            val = b'\x0a' + _varint(len(t)) + t + b'\x15' + struct.pack('<f', float(v))    # tag=1, simple_value=2
            summary += b'\x0a' + _varint(len(val)) + val                                     # Summary.value = 1
        self._f.write(_record(_event(time.time(), step, b'\x2a' + _varint(len(summary)) + summary)))  # summary = 5

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path: str) -> List[Tuple[int, Dict[str, float]]]:
    """Parses an event file back (verifying both CRCs of every record) -> [(step, {tag: value})]."""
    buf = open(path, 'rb').read()
    pos, out = 0, []
    while pos < len(buf):
        n = struct.unpack_from('<Q', buf, pos)[0]
        if unmask_crc(struct.unpack_from('<I', buf, pos + 8)[0]) != crc32c(buf[pos:pos + 8]):
            raise ValueError('event record length checksum mismatch')
        data = buf[pos + 12:pos + 12 + n]
        if unmask_crc(struct.unpack_from('<I', buf, pos + 12 + n)[0]) != crc32c(data):
            raise ValueError('event record data checksum mismatch')
        pos += 16 + n
        step, scal = 0, {}
        for f, _, v in _pb_fields(data):
            if f == 2:
                step = v
            elif f == 5:
                for f2, _, val in _pb_fields(v):
                    if f2 == 1:
                        tag, sv = None, None
                        for f3, wt, x in _pb_fields(val):
                            if f3 == 1:
                                tag = x.decode()
                            elif f3 == 2 and wt == 5:
                                sv = struct.unpack('<f', struct.pack('<I', x))[0]
                        if tag is not None and sv is not None:
                            scal[tag] = sv
        out.append((step, scal))
    return out
