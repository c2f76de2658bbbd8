"""Cross-mode consistency at configurations no oracle test runs: every sub-step's gradients in the f32x3 (and bf16) mode against the
exact-fp32 MFMA mode of the same engine, same state, same batch.   python tools/gpu_mode_consistency.py
(f32x3 vs f32: two fp32 evaluations -- 1e-5 ... 1e-3 with kink flips; bf16 vs f32: the bf16 noise floor, 3e-2 ... 3e-1.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import edgegan_amd as ea


PREFIX = {1: ('D/',), 2: ('D_patch2/',), 3: ('D_patch3/',), 4: ('D2/',), 5: ('G1/', 'G2/'), 6: ('E/',), 7: ('G1/', 'G2/')}


def grads(cfg_kw, dtype, values, batch):
    cfg = ea.Config(compute_dtype=dtype, **cfg_kw)
    net = ea.EdgeGAN(cfg, values)
    net.upload_batch(batch['images'], batch['z'], batch['alpha'], batch['eps'])
    out = {}
    for k in range(1, 8):
        if k == 4 and not cfg.multiclasses:
            continue
        net.substep(k, apply=False)
        out[k] = {}
        for sp in net.specs:
            if sp.trainable and sp.name.startswith(PREFIX[k]):
                try:
                    out[k][sp.name] = net.get_variable(sp.name + '/grad').astype(np.float64)
                except Exception:
                    pass
    net.close()
    return out


for kw in [dict(batch_size=33), dict(batch_size=6, height=32, width=64), dict(batch_size=8, height=128, width=256), dict(batch_size=5, num_classes=0),
           dict(batch_size=3)]:
    cfg = ea.Config(**kw)
    values = ea.init_variables(cfg, 7)
    batch = ea.synthetic_batch(cfg, seed=9)
    ref = grads(kw, 'f32', values, batch)
    for mode in ('f32x3', 'bf16'):
        got = grads(kw, mode, values, batch)
        line = []
        for k in sorted(ref):
            errs = []
            for n, g in ref[k].items():
                h = got[k].get(n)
                if h is None or not np.any(g):
                    continue
                gm = max(np.abs(v).max() for v in ref[k].values())
                if np.abs(g).max() < 1e-9 * gm:
                    continue
                errs.append(np.linalg.norm(h - g) / (np.linalg.norm(g) + 1e-5 * gm * np.sqrt(g.size)))
            line.append('%d: med %.1e max %.1e (%d)' % (k, np.median(errs), max(errs), len(errs)))
        print(kw, mode, 'vs f32 |', ' | '.join(line), flush=True)
