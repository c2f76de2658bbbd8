import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import edgegan_amd as ea, edgegan_oracle as O, parity

B = int(os.environ.get('B', '4'))
cfg = ea.Config(batch_size=B, num_classes=0)
values = ea.init_variables(cfg, 5)
batch = ea.synthetic_batch(cfg, seed=99)
net = ea.EdgeGAN(cfg, values)
net.upload_batch(**batch)
om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
inp = O.StepInputs(*[parity.t64(batch[k]) for k in ('images', 'z', 'alpha', 'eps')])
grads, losses = om.compute_grads('e_optim', inp, inp.eps[0])
net.substep(6, False)
for vn, g in grads.items():
    h = net.get_variable(vn + '/grad').astype(np.float64)
    r = g.numpy()
    d = np.abs(h - r)
    scale = np.abs(r).max()
    bad = d > 1e-4 * scale
    msg = f'{vn}: relerr {d.max()/scale:.2e}, bad frac {bad.mean():.4f}'
    if bad.any() and r.ndim == 4:
        idx = np.argwhere(bad)
        msg += f' | bad taps(kh) {sorted(set(idx[:,0]))} (kw) {sorted(set(idx[:,1]))} cin-range [{idx[:,2].min()},{idx[:,2].max()}] cout-range [{idx[:,3].min()},{idx[:,3].max()}]'
        # is the error a scaled copy?
        ratio = (h[bad] / (r[bad] + 1e-300))
        msg += f' ratio med {np.median(ratio):.4f}'
    print(msg)
