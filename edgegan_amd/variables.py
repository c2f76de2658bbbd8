"""Reference initialisers for the engine's variable registry."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ._lib import Config, variable_specs


def _truncated_normal(rng: np.random.RandomState, shape, std: float) -> np.ndarray:
    """tf.truncated_normal_initializer: resample values beyond 2 sigma (conv.py:20-22)."""
    x = rng.normal(0.0, 1.0, size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * std


def init_variables(cfg: Config, seed: int = 0) -> Dict[str, np.ndarray]:
    """One float32 array per variable, drawn as the reference's initialisers would:
    conv2d w truncated normal 0.02 (conv.py:20-22); deconv2d w / linear Matrix / mlp w / classifier
    convs normal 0.02 (conv.py:46-48, linear.py:15-18,83-84, classifier.py:44); fully_connected
    xavier-uniform (linear.py:36); spectral-norm u truncated normal 1.0 (normalization.py:43-44);
    constants for biases (0), update_gate bias (0.5), PReLU (0.2), BN gamma (1)."""
    rng = np.random.RandomState(seed)
    out = {}
    for s in variable_specs(cfg):
        shape = s.shape
        if s.init_kind == 0:
            a = np.full(shape, s.init_param, dtype=np.float64)
        elif s.init_kind == 1:
            a = rng.normal(0.0, s.init_param, size=shape)
        elif s.init_kind == 2:
            a = _truncated_normal(rng, shape, s.init_param)
        elif s.init_kind == 3:
            fan_in, fan_out = shape[0], shape[1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, size=shape)
        else:
            raise ValueError(s.init_kind)
        out[s.name] = np.array(a, dtype=np.float32, order='C').reshape(shape)
    return out
