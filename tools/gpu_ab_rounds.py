"""Two BUILDS of the library (e.g. the previous round's HEAD and this one) timed alternately in one process group on one
box, through the core entry points only (create / upload / train_step / sync), so that an older library without the newer
exports still loads:   python tools/gpu_ab_rounds.py libA.so libB.so [libC.so ...] [--rounds R] [--height H --width W --batch B --dtype D]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import ctypes, time
    sys.path.insert(0, ROOT)

    class TolerantCDLL(ctypes.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if name.startswith('eg_'):
                    class _Missing:
                        argtypes = None
                        restype = None
                    m = _Missing()
                    self.__dict__[name] = m
                    return m
                raise
    import edgegan_amd._lib as L
    L.C.CDLL = TolerantCDLL
    import edgegan_amd as ea
    H, W, B, dtype, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
    cfg = ea.Config(batch_size=B, height=H, width=W, compute_dtype=dtype)
    net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
    b = ea.synthetic_batch(cfg, seed=1234)
    net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
    for _ in range(2):
        net.train_step()
    net.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.train_step()
    net.sync()
    print('IPS %.2f' % (B * steps / (time.perf_counter() - t0)))
    net.close()
    sys.exit(0)
import argparse
ap = argparse.ArgumentParser()
ap.add_argument('libs', nargs='+')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--height', type=int, default=64)
ap.add_argument('--width', type=int, default=128)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--dtype', default='f32x3')
ap.add_argument('--steps', type=int, default=8)
a = ap.parse_args()
res = {l: [] for l in a.libs}
for r in range(a.rounds):
    for l in a.libs:
        env = dict(os.environ, EG_LIB_PATH=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, __file__, 'child', str(a.height), str(a.width), str(a.batch), a.dtype, str(a.steps)],
                             env=env, capture_output=True, text=True)
        v = [ln for ln in out.stdout.splitlines() if ln.startswith('IPS ')]
        if not v:
            print(l, 'FAILED', out.stderr[-1500:])
            sys.exit(1)
        res[l].append(float(v[-1][4:]))
for l in a.libs:
    print(l, ['%.1f' % v for v in res[l]], 'median %.1f' % sorted(res[l])[len(res[l]) // 2])
