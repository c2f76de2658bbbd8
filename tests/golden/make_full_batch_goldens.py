#!/usr/bin/env python
"""Writes tests/golden/full_batch.npz: digests of the oracle's full-size sub-step gradients (tests/full_batch_lib.py says what a
digest is and why).  Every case is one evaluation of oracle/edgegan_oracle.py -- fp64 autograd of the sub-step's loss at the
given state (edgegan/models/edgegan.py:109-124), plus the oracle's own fp32 run for the per-variable self-deviation, under
O.kink_recorder() and with the PReLU slope conditioning recorded; the bf16 cases run both under O.bf16_operands().

usage: python tests/golden/make_full_batch_goldens.py [case-key ...]     (no argument: every case; existing entries of other
cases are kept).  Takes minutes per case on 8 cores (B = 64 in fp64)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import edgegan_amd as ea          # noqa: E402
import edgegan_oracle as O        # noqa: E402
import full_batch_lib as F        # noqa: E402
import parity                     # noqa: E402

MARGIN, SMALL_EXTENT = 3e-6, 256      # tests/test_tight_gpu.py / tests/golden/select_fixtures.py
THREAD_SAMPLES = (1, 2, 4, 8)


def evaluate(case):
    """-> (g64 {vn: np fp64}, losses {k: float}, self_dev {vn: float}, kappa {vn: float}, near_kink)"""
    key, name, kw, vseed, bseed, bf16 = case
    cfg = ea.Config(**kw)
    values = ea.init_variables(cfg, vseed)
    b = ea.synthetic_batch(cfg, seed=bseed)
    with O.bf16_operands(bf16):
        om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
        inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
        om32 = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float32)
        inp32 = O.StepInputs(*[torch.tensor(np.asarray(b[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
        # The oracle's own fp32 evaluation is ONE sample of fp32 noise per summation order -- which side of a ReLU / PReLU /
        # min-max kink a pre-activation within rounding of it lands on differs from sample to sample (tests/test_tight_gpu.py) --
        # and torch's order depends on the thread count: four samples (1, 2, 4, 8 threads), the per-variable maximum is kept.
        g32s = []
        nthreads = torch.get_num_threads()
        for nt in THREAD_SAMPLES:
            torch.set_num_threads(nt)
            g32s.append(om32.compute_grads(name, inp32, inp32.eps[0])[0])
        torch.set_num_threads(nthreads)
        O._SLOPE_TERMS = {}
        try:
            with O.kink_recorder() as rec:
                g64, l64 = om.compute_grads(name, inp, inp.eps[0])
        finally:
            O._SLOPE_TERMS = None
    near_kink = rec.margin(SMALL_EXTENT) < MARGIN
    gmax = max(float(g.abs().max()) for g in g64.values())
    self_dev = {vn: max(parity.rel_l2(g32[vn].double().numpy(), g.numpy(), 1e-5 * gmax) for g32 in g32s) for vn, g in g64.items()}
    kappa = {vn: l1 / max(abs(float(g64[vn])), 1e-300) for vn, l1 in getattr(om, 'last_slope_l1', {}).items() if vn in g64}
    return ({vn: g.numpy() for vn, g in g64.items()}, {k: float(v) for k, v in l64.items()}, self_dev, kappa, near_kink, gmax)


def entries(case):
    g64, losses, self_dev, kappa, near_kink, gmax = evaluate(case)
    pre = case[0] + '|'
    out = {pre + 'gmax': np.array([gmax]), pre + 'near_kink': np.array([1.0 if near_kink else 0.0])}
    for vn, g in g64.items():
        for part, arr in F.digest(vn, g).items():
            out[f'{pre}v|{vn}|{part}'] = arr
        out[f'{pre}self_dev|{vn}'] = np.array([self_dev[vn]])
        if vn in kappa:
            out[f'{pre}kappa|{vn}'] = np.array([kappa[vn]])
    for k, v in losses.items():
        out[f'{pre}loss|{k}'] = np.array([v])
    return out


if __name__ == '__main__':
    want = sys.argv[1:] or [c[0] for c in F.CASES]
    for key in want:
        new = entries(F.CASE[key])
        have = dict(np.load(F.GOLDEN)) if os.path.exists(F.GOLDEN) else {}      # (re-read: several cases may be generated in parallel)
        have = {k: v for k, v in have.items() if not k.startswith(key + '|')}
        have.update(new)
        tmp = F.GOLDEN + f'.tmp{os.getpid()}.npz'
        np.savez_compressed(tmp, **have)
        os.replace(tmp, F.GOLDEN)
        print(key, len(new), 'entries', flush=True)
