"""Sub-step gradient errors (batch 3 fixture of tests/test_step_gpu.py::test_substeps_odd_batch) under the fp32 implementations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import edgegan_amd as ea
import parity

cfg = ea.Config(batch_size=int(os.environ.get('B', '3')))
values = ea.init_variables(cfg, 11)
net = ea.EdgeGAN(cfg, values)
rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=3))
sel = os.environ.get('SEL', 'substep 4')
rows = [r for r in rows if sel in r[0] and ' grad ' in r[0]]
for r in rows:
    print(f'{r[1]:10.3e} tol {r[2]:7.0e}  {r[0]}')
