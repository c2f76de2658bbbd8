"""Race check for the concurrent sub-step schedule: N iterations with a fresh batch each, run twice with lanes on and
once with EG_LANES=0 semantics (separate process via env); all three must end on bitwise identical variables."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea

N = int(os.environ.get('ITERS', '200'))
cfg = ea.Config(batch_size=int(os.environ.get('B', '16')), compute_dtype=os.environ.get('DTYPE', 'f32'))
values = ea.init_variables(cfg, 0)
net = ea.EdgeGAN(cfg, values)
for it in range(N):
    b = ea.synthetic_batch(cfg, seed=it)
    net.update_model(b['images'], b['z'], alpha=b['alpha'], eps=b['eps'])
h = hashlib.sha256()
for name, v in sorted(net.get_variables(slots=True).items()):
    h.update(name.encode())
    h.update(np.ascontiguousarray(v).tobytes())
print('EG_LANES=%s iters=%d sha256=%s' % (os.environ.get('EG_LANES', 'default'), N, h.hexdigest()))
net.close()
