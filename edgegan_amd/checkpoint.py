"""Checkpoint save / resume with the reference's layout and rules (edgegan/models/edgegan.py:635-661).

``save``: ``<checkpoint_dir>/EdgeGAN-Model-<step>.{index,data-00000-of-00001}`` + ``checkpoint`` state file, holding
what ``tf.train.Saver()`` with no var_list saves (edgegan.py:421): every global variable -- trainables, the
spectral-norm ``u`` vectors, BatchNorm moving statistics and the RMSProp slots ``<var>/RMSProp`` (rms) and
``<var>/RMSProp_1`` (momentum) of every variable that receives a gradient (the classifier's dead ``disc`` head
has none, quirk N8).
``load``: latest checkpoint of the state file; the resume counter is the trailing integer of its file name
(edgegan.py:651-652).
"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional, Tuple

import numpy as np

from . import tf_bundle

MODEL_NAME = 'EdgeGAN-Model'          # edgegan.py:659-661
DEAD_PREFIX = 'D2/Conv_1/'            # never receives a gradient => TF creates no optimizer slots for it


def checkpoint_tensors(net) -> Dict[str, np.ndarray]:
    out = {}
    for s in net.specs:
        out[s.name] = net.get_variable(s.name, s.shape)
        if s.trainable and not s.name.startswith(DEAD_PREFIX):
            out[s.name + '/RMSProp'] = net.get_variable(s.name + '/RMSProp', s.shape)
            out[s.name + '/RMSProp_1'] = net.get_variable(s.name + '/RMSProp_1', s.shape)
    return out


def save(net, checkpoint_dir: str, step: int) -> str:
    os.makedirs(checkpoint_dir, exist_ok=True)
    name = f'{MODEL_NAME}-{int(step)}'
    prefix = os.path.join(checkpoint_dir, name)
    print(' [*] Saving checkpoints...')
    tf_bundle.write_bundle(prefix, checkpoint_tensors(net))
    _, allp = tf_bundle.read_checkpoint_state(checkpoint_dir)
    allp = [p for p in allp if p != name] + [name]
    allp = allp[-5:]                                   # Saver(max_to_keep=5) default
    tf_bundle.write_checkpoint_state(checkpoint_dir, name, allp)
    return prefix


def _match_name(want: str, have: Dict[str, np.ndarray]) -> Optional[str]:
    if want in have:
        return want
    # The nested scope of the spectral-norm `u` (normalization.py:42-44 re-enters the variable's scope inside
    # the current one) is the one name this build could not confirm against TF: also accept the un-nested form
    # 'D2/<scope>/u' for the expected 'D2/<scope>/D2/<scope>/u'.
    if want.endswith('/u'):
        parts = want[:-2].split('/')
        half = len(parts) // 2
        if len(parts) % 2 == 0 and parts[:half] == parts[half:]:
            alt = '/'.join(parts[:half]) + '/u'
            if alt in have:
                return alt
    return None


def load(net, checkpoint_dir: str, strict: bool = True) -> Tuple[bool, int]:
    print(' [*] Reading checkpoints {}...'.format(checkpoint_dir))
    latest, _ = tf_bundle.read_checkpoint_state(checkpoint_dir)
    if not latest:
        print(' [*] Failed to find a checkpoint')
        return False, 0
    ckpt_name = os.path.basename(latest)
    have = tf_bundle.read_bundle(os.path.join(checkpoint_dir, ckpt_name))
    missing = []
    for s in net.specs:
        k = _match_name(s.name, have)
        if k is None or tuple(have[k].shape) != tuple(s.shape):
            missing.append(s.name if k is None else f'{s.name} (shape {tuple(have[k].shape)} != {tuple(s.shape)})')
            continue
        net.set_variable(s.name, have[k].astype(np.float32))
        for slot in ('/RMSProp', '/RMSProp_1'):
            if s.trainable and s.name + slot in have:
                if tuple(have[s.name + slot].shape) != tuple(s.shape):
                    missing.append(f'{s.name + slot} (shape {tuple(have[s.name + slot].shape)} != {tuple(s.shape)})')
                    continue
                net.set_variable(s.name + slot, have[s.name + slot].astype(np.float32))
    if missing and strict:
        raise KeyError(f'checkpoint {ckpt_name} lacks or mis-shapes {len(missing)} variables, e.g. {missing[:5]}')
    if missing:
        print(f' [!] {len(missing)} variables not found in the checkpoint (kept at their current values), '
              f'e.g. {missing[:3]}')
    counter = int(next(re.finditer(r'(\d+)(?!.*\d)', ckpt_name)).group(0))
    print(' [*] Success to read {}'.format(ckpt_name))
    return True, counter
