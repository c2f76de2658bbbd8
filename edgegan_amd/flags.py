"""tf.app.flags-compatible command lines of the reference (edgegan/train.py:13-75, edgegan/test.py:17-65).

Accepted syntax is absl's: ``--name=value``, ``--name value``, booleans as ``--flag`` / ``--noflag`` /
``--flag=true|false``.  Flags the reference defines but never reads (test.py's ``type``, ``optim``, ``model``) are
accepted for command-line compatibility; flags selecting graph variants outside the hot path
(``if_resnet_g/d``, ``if_resnet_e=False``) are rejected loudly; E_norm / G_norm / D_norm take 'instance' or 'batch'."""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Tuple

_COMMON = [
    ('gpu', str, '0'), ('name', str, 'edgegan'), ('outputsroot', str, 'outputs'), ('epoch', int, 100),
    ('learning_rate', float, 0.0002), ('train_size', float, math.inf), ('batch_size', int, 64),
    ('input_height', int, 64), ('input_width', int, 128), ('output_height', int, 64), ('output_width', int, 128),
    ('dataset', str, 'class14'), ('input_fname_pattern', str, '*png'), ('checkpoint_dir', str, None),
    ('logdir', str, None), ('dataroot', str, './data'), ('crop', bool, False), ('multiclasses', bool, True),
    ('num_classes', int, 14), ('if_resnet_e', bool, True), ('if_resnet_g', bool, False), ('if_resnet_d', bool, False),
    ('E_norm', str, 'instance'), ('G_norm', str, 'instance'), ('D_norm', str, 'instance'),
]
TRAIN_FLAGS = _COMMON + [
    ('save_checkpoint_frequency', int, 500), ('stage1_zl_loss', float, 10.0),
    ('SPECTRAL_NORM_UPDATE_OPS', str, 'spectral_norm_update_ops'), ('lambda_gp', float, 10.0),
    ('use_image_discriminator', bool, True), ('image_dis_size', int, 128), ('use_edge_discriminator', bool, True),
    ('edge_dis_size', int, 128), ('joint_dweight', float, 1.0), ('image_dweight', float, 1.0),
    ('edge_dweight', float, 1.0), ('z_dim', int, 100),
    # not a reference flag.  Default (on): what the reference prints and logs -- all losses re-evaluated AFTER the update with
    # fresh alpha / eps (edgegan.py:427-433,464-486, incl. the 6 extra BatchNorm moving-average updates per iteration) and the
    # histogram / image summaries: the same numbers a user of the reference sees.  --noreference_evaluation = the fast loop:
    # losses captured inside the step, scalars only, no extra passes, iterations pipelined (python bench.py --mode loop)
    ('reference_evaluation', bool, True),
]
TEST_FLAGS = _COMMON + [
    ('test_output_dir', str, 'test_output'), ('output_combination', str, 'full'), ('type', str, 'gpwgan'),
    ('optim', str, 'rmsprop'), ('model', str, 'old'),
]


def _to_bool(v: str) -> bool:
    if v.lower() in ('true', 't', '1', 'yes'):
        return True
    if v.lower() in ('false', 'f', '0', 'no'):
        return False
    raise ValueError(f'not a boolean: {v!r}')


def parse(argv: List[str], defs: List[Tuple[str, type, object]]) -> SimpleNamespace:
    table: Dict[str, Tuple[type, object]] = {n: (t, d) for n, t, d in defs}
    vals = {n: d for n, t, d in defs}
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if not a.startswith('-'):
            raise SystemExit(f'unexpected argument {a!r}')
        a = a.lstrip('-')
        name, eq, val = a.partition('=')
        if name not in table and name.startswith('no') and name[2:] in table and table[name[2:]][0] is bool:
            if eq:
                raise SystemExit(f'--{name} takes no value')
            vals[name[2:]] = False
            continue
        if name not in table:
            raise SystemExit(f'unknown flag --{name}')
        t = table[name][0]
        if t is bool:
            if eq:
                vals[name] = _to_bool(val)
            elif i < len(argv) and argv[i].lower() in ('true', 'false'):
                vals[name] = _to_bool(argv[i])
                i += 1
            else:
                vals[name] = True
            continue
        if not eq:
            if i >= len(argv):
                raise SystemExit(f'flag --{name} needs a value')
            val = argv[i]
            i += 1
        vals[name] = t(val)
    return SimpleNamespace(**vals)


def check_supported(flags: SimpleNamespace):
    """The engine builds the default graph of the reference (SURVEY.md 2.1 scope column)."""
    bad = []
    if flags.if_resnet_g or flags.if_resnet_d or not flags.if_resnet_e:
        bad.append('if_resnet_g / if_resnet_d / if_resnet_e must keep their defaults (False, False, True)')
    for nm in ('E_norm', 'G_norm', 'D_norm'):   # normalization.py:12: 'instance' | 'batch' (None is not reachable from a string flag)
        if getattr(flags, nm) not in ('instance', 'batch'):
            bad.append(f'{nm} must be "instance" or "batch"')
    if bad:
        raise SystemExit('unsupported flag combination:\n  ' + '\n  '.join(bad))
