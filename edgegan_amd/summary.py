"""TensorBoard event files for the summaries of the reference (edgegan/models/edgegan.py:344-411), without TensorFlow:
scalars (``tf.summary.scalar``), histograms (``tf.summary.histogram``: ``HistogramProto`` with TensorFlow's default
bucket limits, empty runs collapsed as ``histogram::Histogram::EncodeToProto`` does) and images
(``tf.summary.image``: up to 3 images per tag as ``<tag>/image/<i>``, floats normalised as
``summary_image_op.cc:NormalizeFloatImage`` does, PNG-encoded).  The reference spends two extra forward passes per
iteration on ``sess.run(d_sum / g_sum)`` (edgegan.py:427-433); the scalars here come out of the training step itself
(``eg_read_losses``) or, with the reference's post-update semantics, out of one ``eg_eval_losses`` call that also
returns the tensors behind the histogram / image summaries.

File format (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/util/event.proto): a sequence of records
``uint64 length | masked crc32c(length) | data | masked crc32c(data)`` whose data are ``Event`` protos;
the first event carries ``file_version = "brain.Event:2"``."""
from __future__ import annotations

import os
import socket
import struct
import time
import io
from typing import Dict, Iterable, List, Tuple

import numpy as np

from .tf_bundle import crc32c, mask_crc, unmask_crc, _varint, _pb_fields

# tags exactly as the reference names them (d_sum and g_sum members)
SCALAR_TAGS = ['joint_dis_dloss', 'loss_d_ac', 'image_dis_dloss', 'edge_dis_dloss',          # d_sum
               'zl_loss', 'joint_dis_gloss', 'loss_g_ac', 'edge_gloss', 'image_gloss',       # g_sum
               'image_dis_gloss', 'edge_dis_gloss']


# histogram / image members of d_sum and g_sum (edgegan.py:345-411) -> key of EdgeGAN.evaluate(tensors=True)
HISTOGRAM_TAGS = {'d': ('d_logits', 0), 'd_': ('d_logits', 1), 'imageD': ('d_logits', 2), 'imageDfake': ('d_logits', 3),
                  'edgeD': ('d_logits', 4), 'edgeDfake': ('d_logits', 5)}      # + 'z' (the feed itself)
IMAGE_TAGS = {'G1': 'edge_output', 'G2': 'image_output', 'resized_inputs_image': 'resized_inputs',
              'resized_G_image': 'resized_image_output', 'resized_inputs_p3_image': 'resized_edges',
              'resized_G_p3_image': 'resized_edge_output'}                      # + 'inputs' (the feed itself)

_BUCKET_LIMITS = None


def default_bucket_limits() -> np.ndarray:
    """tensorflow/core/lib/histogram/histogram.cc InitDefaultBucketsInner: +-(1e-12 * 1.1^k) up to 1e20, 0, DBL_MAX."""
    global _BUCKET_LIMITS
    if _BUCKET_LIMITS is None:
        pos, v = [], 1e-12
        while v < 1e20:
            pos.append(v)
            v *= 1.1
        pos.append(np.finfo(np.float64).max)
        _BUCKET_LIMITS = np.array([-x for x in reversed(pos)] + [0.0] + pos, dtype=np.float64)
    return _BUCKET_LIMITS


def _pb_double(field: int, v: float) -> bytes:
    return bytes([(field << 3) | 1]) + struct.pack('<d', float(v))


def _pb_bytes(field: int, b: bytes) -> bytes:
    return bytes([(field << 3) | 2]) + _varint(len(b)) + b


def histogram_proto(values) -> bytes:
    """HistogramProto (min=1, max=2, num=3, sum=4, sum_squares=5, packed bucket_limit=6, packed bucket=7)."""
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    limits = default_bucket_limits()
    idx = np.searchsorted(limits, v, side='right')          # std::upper_bound: first limit > value
    counts = np.bincount(np.minimum(idx, len(limits) - 1), minlength=len(limits)).astype(np.float64)
    lim_out, cnt_out, i = [], [], 0
    while i < len(limits):                                   # EncodeToProto: collapse runs of empty buckets
        end, cnt = limits[i], counts[i]
        i += 1
        if cnt <= 0.0:
            while i < len(limits) and counts[i] <= 0.0:
                end, cnt = limits[i], counts[i]
                i += 1
        lim_out.append(end)
        cnt_out.append(cnt)
    return (_pb_double(1, v.min() if v.size else 0.0) + _pb_double(2, v.max() if v.size else 0.0) + _pb_double(3, v.size) +
            _pb_double(4, v.sum()) + _pb_double(5, (v * v).sum()) +
            _pb_bytes(6, struct.pack(f'<{len(lim_out)}d', *lim_out)) + _pb_bytes(7, struct.pack(f'<{len(cnt_out)}d', *cnt_out)))


def normalize_float_image(img: np.ndarray) -> np.ndarray:
    """summary_image_op.cc NormalizeFloatImage: any negative value -> 0 maps to 128 and the largest magnitude to
    +-127 around it; otherwise the maximum maps to 255."""
    a = np.asarray(img, dtype=np.float32)
    fin = a[np.isfinite(a)]
    lo, hi = (float(fin.min()), float(fin.max())) if fin.size else (0.0, 0.0)
    if lo < 0:
        m = max(abs(lo), abs(hi))
        scale, offset = (127.0 / m if m > 0 else 0.0), 128.0
    else:
        scale, offset = (255.0 / hi if hi > 0 else 0.0), 0.0
    out = np.where(np.isfinite(a), a * scale + offset, 0.0)
    return np.clip(out, 0, 255).astype(np.uint8)


def image_proto(img_u8: np.ndarray) -> bytes:
    """Summary.Image (height=1, width=2, colorspace=3, encoded_image_string=4), PNG."""
    from PIL import Image
    h, w, ch = img_u8.shape
    buf = io.BytesIO()
    Image.fromarray(img_u8[:, :, 0] if ch == 1 else img_u8).save(buf, format='PNG')
    return b'\x08' + _varint(h) + b'\x10' + _varint(w) + b'\x18' + _varint(ch) + _pb_bytes(4, buf.getvalue())


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

    def _write_values(self, vals: List[bytes], step: int):
        summary = b''.join(_pb_bytes(1, v) for v in vals)                                    # Summary.value = 1
        self._f.write(_record(_event(time.time(), step, _pb_bytes(5, summary))))             # Event.summary = 5

    def add_histograms(self, values: Dict[str, np.ndarray], step: int):
        """tf.summary.histogram members of d_sum / g_sum (edgegan.py:345,371-372,376-379,392-395)."""
        self._write_values([_pb_bytes(1, t.encode()) + _pb_bytes(5, histogram_proto(v)) for t, v in values.items()], step)

    def add_images(self, values: Dict[str, np.ndarray], step: int, max_outputs: int = 3):
        """tf.summary.image members (edgegan.py:346-349,380-383,396-399): [B,H,W,C] float -> <tag>/image/<i>."""
        vals = []
        for tag, batch in values.items():
            for i in range(min(max_outputs, batch.shape[0])):
                name = f'{tag}/image/{i}' if max_outputs > 1 else f'{tag}/image'
                vals.append(_pb_bytes(1, name.encode()) + _pb_bytes(4, image_proto(normalize_float_image(batch[i]))))
        self._write_values(vals, step)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_event_tags(path: str) -> List[Tuple[int, Dict[str, str]]]:
    """[(step, {tag: 'scalar' | 'histo' | 'image'})] -- for tests."""
    buf = open(path, 'rb').read()
    pos, out = 0, []
    while pos < len(buf):
        n = struct.unpack_from('<Q', buf, pos)[0]
        data = buf[pos + 12:pos + 12 + n]
        pos += 16 + n
        step, tags = 0, {}
        for f, _, v in _pb_fields(data):
            if f == 2:
                step = v
            elif f == 5:
                for f2, _, val in _pb_fields(v):
                    if f2 == 1:
                        tag, kind = None, None
                        for f3, wt, x in _pb_fields(val):
                            if f3 == 1:
                                tag = x.decode()
                            elif f3 in (2, 4, 5):
                                kind = {2: 'scalar', 4: 'image', 5: 'histo'}[f3]
                        if tag is not None:
                            tags[tag] = kind
        out.append((step, tags))
    return out


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
