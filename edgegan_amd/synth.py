"""Deterministic synthetic feeds of the shape the reference trains on."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ._lib import Config


def synthetic_batch(cfg: Config, seed: int = 1234, rank: int = 0) -> Dict[str, np.ndarray]:
    """images float32 [B,H,W,3] in [-1,1]: left half an 'edge map' (white with ~6 % dark stroke pixels),
    right half a 'photo' (random blob on white covering ~50 %), matching the statistics of the
    reference's images/dataset_example PNGs; z ~ N(0,1) [B,z_dim] (dataset.py:72-73) with the class id
    `b mod num_classes` appended (dataset.py:82-84); alpha ~ U(0,1) [3,B]; eps ~ N(0,1)."""
    rng = np.random.RandomState(seed + 7919 * rank)
    B, H, W = cfg.batch_size, cfg.height, cfg.width
    img = np.ones((B, H, W, cfg.c_dim), dtype=np.float32)
    half = W // 2
    strokes = rng.uniform(size=(B, H, half, 1)) < 0.06
    img[:, :, :half, :] = np.where(strokes, rng.uniform(-1.0, 0.0, size=(B, H, half, 1)), 1.0)
    yy, xx = np.mgrid[0:H, 0:half]
    for b in range(B):
        cy, cx = rng.uniform(0.3, 0.7) * H, rng.uniform(0.3, 0.7) * half
        ry, rx = rng.uniform(0.25, 0.45) * H, rng.uniform(0.25, 0.45) * half
        mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        photo = rng.uniform(-1.0, 1.0, size=(H, half, cfg.c_dim))
        img[b, :, half:, :] = np.where(mask[..., None], photo, 1.0)
    z = rng.normal(0.0, 1.0, size=(B, cfg.z_dim)).astype(np.float32)
    if cfg.multiclasses:
        cls = (np.arange(B) % cfg.num_classes).astype(np.float32)[:, None]
        z = np.concatenate([z, cls], axis=1)
    alpha = rng.uniform(0.0, 1.0, size=(3, B)).astype(np.float32)
    eps = rng.normal(0.0, 1.0, size=(4,)).astype(np.float32)
    return {'images': np.ascontiguousarray(img, dtype=np.float32), 'z': np.ascontiguousarray(z, dtype=np.float32),
            'alpha': alpha, 'eps': eps}
