"""Is there a state at which the bf16 mode's encoder sub-step CAN be compared with the emulating oracle?
At random init the oracle's own fp32 evaluation of the emulated e_optim deviates from its fp64 evaluation by 0.2 ... 0.3
(tests/test_bf16_gpu.py) -- a tolerance derived from that proves nothing.  This trains the fp32 oracle for N iterations at a small
batch and measures the same self-consistency at the resulting state (CPU only, minutes).
usage: python tools/cpu_eoptim_conditioning.py [iterations=50] [train_batch=4] [eval_batch=16] [out.npz]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch

import edgegan_amd as ea
import edgegan_oracle as O
import parity


def self_dev(values, cfg, b, name, bf16):
    with O.bf16_operands(bf16):
        om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
        inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
        om32 = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float32)
        inp32 = O.StepInputs(*[torch.tensor(np.asarray(b[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
        g64, _ = om.compute_grads(name, inp, inp.eps[0])
        g32, _ = om32.compute_grads(name, inp32, inp32.eps[0])
    gmax = max(float(g.abs().max()) for g in g64.values())
    d = {vn: parity.rel_l2(g32[vn].double().numpy(), g.numpy(), 1e-5 * gmax) for vn, g in g64.items()}
    v = np.array(list(d.values()))
    return float(np.median(v)), float(v.max()), d


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    tb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    eb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    out = sys.argv[4] if len(sys.argv) > 4 else None
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    cfg_t = ea.Config(batch_size=tb)
    values = ea.init_variables(cfg_t, 5)
    cfg_e = ea.Config(batch_size=eb)
    be = ea.synthetic_batch(cfg_e, seed=21)
    for tag, bf in (('emulated bf16', True), ('fp32', False)):
        med, mx, _ = self_dev(values, cfg_e, be, 'e_optim', bf)
        print(f'iteration 0, {tag}: e_optim oracle fp32-vs-fp64 self-deviation median {med:.2e} max {mx:.2e}', flush=True)
    om = O.OracleModel(parity.oracle_config(cfg_t), values, dtype=torch.float32)
    t0 = time.time()
    for it in range(1, iters + 1):
        b = ea.synthetic_batch(cfg_t, seed=1000 + it)
        inp = O.StepInputs(*[torch.tensor(np.asarray(b[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
        om.update_model(inp)
        if it in (10, 25, 50, 100, 200) or it == iters:
            vals = {k: v.astype(np.float32) for k, v in om.numpy_params().items()}
            for tag, bf in (('emulated bf16', True), ('fp32', False)):
                med, mx, _ = self_dev(vals, cfg_e, be, 'e_optim', bf)
                print(f'iteration {it} ({time.time() - t0:.0f} s), {tag}: e_optim self-deviation median {med:.2e} max {mx:.2e}', flush=True)
            for name in ('g_optim_u', 'd_optim'):
                med, mx, _ = self_dev(vals, cfg_e, be, name, True)
                print(f'iteration {it}, emulated bf16: {name} self-deviation median {med:.2e} max {mx:.2e}', flush=True)
    if out:
        np.savez_compressed(out, **{k: v.astype(np.float32) for k, v in om.numpy_params().items()})


if __name__ == '__main__':
    main()
