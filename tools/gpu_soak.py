"""Soak: many iterations with fresh noise per iteration; losses stay finite, arena stops growing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea
cfg = ea.Config(batch_size=int(os.environ.get('B', '16')), height=int(os.environ.get('H', '64')),
                width=int(os.environ.get('W', '128')))
net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
rng = np.random.RandomState(0)
base = ea.synthetic_batch(cfg)
mem = None
t0 = time.time()
N = int(os.environ.get('ITERS', '150'))
for it in range(N):
    b = ea.synthetic_batch(cfg, seed=it)
    net.update_model(b['images'], b['z'])
    if it % 25 == 0 or it == N - 1:
        l = net.losses()
        m = net.memory_stats()
        ok = all(np.isfinite(v) for v in l.values())
        print(it, 'finite' if ok else 'NON-FINITE', 'D %.3f G %.3f zl %.3f cls %.3f' % (
            l['joint_dis_dloss'] + l['image_dis_dloss'] + l['edge_dis_dloss'], l['edge_gloss'] + l['image_gloss'],
            l['zl_loss'], l['loss_d_ac']), 'arena %.0f MB' % (m['arena_reserved_bytes'] / 1e6))
        assert ok
        if it >= 25:
            assert mem is None or m['arena_reserved_bytes'] == mem, 'arena grew after warm-up'
            mem = m['arena_reserved_bytes']
print('soak ok, %.1f s' % (time.time() - t0))
