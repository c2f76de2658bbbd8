import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea
cfg = ea.Config(batch_size=int(os.environ.get('B', '64')))
net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
net.upload_batch(**ea.synthetic_batch(cfg))
for _ in range(2):
    net.train_step()
net.sync()
tot = np.zeros(8)
R = 5
for _ in range(R):
    for k in range(1, 8):
        net.sync(); t = time.perf_counter()
        net.substep(k, True)
        net.sync(); tot[k] += time.perf_counter() - t
names = ['', 'd_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim', 'g_optim_b']
for k in range(1, 8):
    print(f'substep {k} {names[k]:16s} {tot[k] / R * 1e3:7.2f} ms')
print('sum', tot.sum() / R * 1e3)
