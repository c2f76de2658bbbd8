#!/usr/bin/env python
"""Writes the committed golden vectors (tests/golden/*.npz) from the fp64 oracle.   python tests/golden/make_goldens.py

SURVEY.md 8c: (i) per-op forward/backward on tiny shapes incl. odd sizes and borders, (ii) each network forward at
B=2 with PRNG-regenerated weights, (iii) each of the 7 sub-steps: losses, per-variable gradients and the
post-RMSProp weights after 1 and 2 iterations with alpha / eps / z / images fixed (14-class B=2 and single-class
B=4 = BASELINE configs[0]), (iv) inference E -> G1, G2 on the reference's example test PNGs, (v) bicubic / SN /
SAME-pad known-answer vectors.  Case definitions and compute functions: tests/golden_lib.py.
The goldens freeze the ORACLE (they cannot pin it to the reference: no TF here -- "parity unpinned")."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import edgegan_amd as ea          # noqa: E402  (variable registry + initialisers + synthetic feeds; no GPU needed)
import golden_lib as G            # noqa: E402
import parity                     # noqa: E402

# fixtures: (file, config kwargs, variable seed, batch seed)
STEP_FIXTURES = [('steps_14class_b2.npz', dict(batch_size=2), 0, 1234),
                 ('steps_1class_b4.npz', dict(batch_size=4, num_classes=0), 5, 99)]
NET_FIXTURES = [('nets_14class_b2.npz', dict(batch_size=2), 1), ('nets_1class_b2.npz', dict(batch_size=2, num_classes=0), 1)]
INFER_FIXTURE = ('infer_examples.npz', dict(batch_size=1), 7)


def example_inputs():
    """The 4 example test sketches through the product's data pipeline (class id = directory name)."""
    from edgegan_amd import data as D
    dcfg = {'input_height': 64, 'input_width': 128, 'output_height': 64, 'output_width': 128, 'crop': False}
    ds = D.Dataset(HERE, 'dataset_example', np.inf, 1, dcfg, None, 'test')
    imgs, cls = [], []
    for i in range(len(ds)):
        im, files = ds[i]
        imgs.append(np.asarray(im, dtype=np.float32))
        cls.append(np.array([int(os.path.basename(os.path.dirname(files[0])))], dtype=np.int64))
    return imgs, cls


def build(which=None):
    out = {}
    if which in (None, 'ops', 'ops.npz'):
        out['ops.npz'] = G.compute_ops()
    for fn, kw, vseed in NET_FIXTURES:
        if which in (None, fn):
            cfg = ea.Config(**kw)
            out[fn] = G.compute_nets(parity.oracle_config(cfg), ea.init_variables(cfg, vseed), G.net_inputs(cfg))
    for fn, kw, vseed, bseed in STEP_FIXTURES:
        if which in (None, fn):
            cfg = ea.Config(**kw)
            out[fn] = G.compute_steps(parity.oracle_config(cfg), ea.init_variables(cfg, vseed),
                                      ea.synthetic_batch(cfg, seed=bseed), iterations=2)
    if which in (None, INFER_FIXTURE[0]):
        cfg = ea.Config(**INFER_FIXTURE[1])
        imgs, cls = example_inputs()
        out[INFER_FIXTURE[0]] = G.compute_infer(parity.oracle_config(cfg), ea.init_variables(cfg, INFER_FIXTURE[2]), imgs, cls)
    return out


if __name__ == '__main__':
    for fn, d in build().items():
        np.savez_compressed(os.path.join(HERE, fn), **d)
        print(fn, len(d), 'entries', os.path.getsize(os.path.join(HERE, fn)), 'bytes')
