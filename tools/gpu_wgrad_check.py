"""Row-halo filter-gradient kernel (wgrad_x3r) against the per-tap kernel (wgrad_x3t, EG_NO_WGRAD_ROWHALO=1) and against a
float64 einsum on the small shapes: both kernels evaluate the same six limb products per element, so they agree to fp32
summation order.  Experiments build (the switch is a debug environment variable)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# NB, H, W, C, N, pad (0 zero SAME, 1 reflect)
SHAPES = [(64, 64, 64, 128, 128, 0), (64, 32, 32, 256, 256, 0), (64, 16, 16, 512, 512, 0), (64, 8, 8, 768, 768, 0),
          (64, 32, 32, 128, 128, 1), (64, 4, 4, 512, 512, 1), (3, 4, 4, 128, 128, 1), (5, 8, 16, 64, 128, 0),
          (2, 16, 8, 192, 132, 1), (1, 128, 128, 64, 256, 0), (7, 4, 8, 128, 256, 0), (64, 64, 64, 128, 256, 0),
          # k = 4, stride 2 (pad field = 40) and the 5x5 stride-2 transposed convolution (pad field = 50: H, W = its INPUT map)
          (128, 32, 32, 128, 256, 40), (128, 16, 16, 256, 512, 40), (64, 8, 16, 256, 512, 40), (128, 64, 64, 64, 128, 40),
          (3, 8, 8, 64, 128, 40), (5, 16, 8, 128, 132, 40), (64, 4, 4, 512, 256, 50), (64, 8, 8, 256, 128, 50), (3, 4, 8, 256, 128, 50)]
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import numpy as np
    import edgegan_amd as ea
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype=os.environ.get('DTYPE', 'f32x3')))
    out = {}
    for (NB, H, W, C, N, pad) in SHAPES:
        rng = np.random.RandomState(NB + C + H)
        x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
        if pad == 50:
            w = (rng.normal(size=(5, 5, N, C)) * 0.05).astype(np.float32)
            b = np.zeros(N, np.float32)
            dy = rng.normal(size=(NB, 2 * H, 2 * W, N)).astype(np.float32)
            args = ('deconv2d', [x, w, b, dy], [NB, H, W, C, N], [dy.shape, x.shape, w.shape, b.shape])
        elif pad == 40:
            w = (rng.normal(size=(4, 4, C, N)) * 0.05).astype(np.float32)
            dy = rng.normal(size=(NB, H // 2, W // 2, N)).astype(np.float32)
            args = ('conv2d', [x, w, None, dy], [NB, H, W, C, N, 4, 2, 0], [dy.shape, x.shape, w.shape, None])
        else:
            w = (rng.normal(size=(3, 3, C, N)) * 0.05).astype(np.float32)
            dy = rng.normal(size=(NB, H, W, N)).astype(np.float32)
            args = ('conv2d', [x, w, None, dy], [NB, H, W, C, N, 3, 1, pad], [(NB, H, W, N), x.shape, w.shape, None])
        net.test_op(*args)
        net.profile_begin()
        dw = net.test_op(*args)[2]
        k = [q for q in net.profile_end()['kernels'] if 'wgrad' in q['name']]
        key = f'{NB}_{H}_{W}_{C}_{N}_{pad}'
        np.save(os.path.join(sys.argv[2], key + '.npy'), dw)
        out[key] = [(q['name'], round(q['ms'] / q['launches'], 4), round(q['flops'] / q['ms'] / 1e9, 1)) for q in k]
    print('KERNELS ' + json.dumps(out))
    net.close()
    sys.exit(0)
import numpy as np, tempfile
dirs = {}
for tag, env in (('rowhalo', {}), ('pertap', {'EG_WGRAD_ROWHALO': '0'})):
    d = tempfile.mkdtemp()
    e = dict(os.environ, EG_LIB_PATH=os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so'), **env)
    r = subprocess.run([sys.executable, __file__, 'child', d], env=e, capture_output=True, text=True)
    if r.returncode != 0:
        print(tag, 'FAILED', r.stderr[-3000:])
        sys.exit(1)
    dirs[tag] = (d, json.loads([l for l in r.stdout.splitlines() if l.startswith('KERNELS ')][-1][8:]))
bad = 0
for (NB, H, W, C, N, pad) in SHAPES:
    key = f'{NB}_{H}_{W}_{C}_{N}_{pad}'
    a = np.load(os.path.join(dirs['rowhalo'][0], key + '.npy'))
    b = np.load(os.path.join(dirs['pertap'][0], key + '.npy'))
    err = float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if np.all(np.isfinite(a)) else float('inf')
    ref = ''
    if pad < 2 and NB * H * W * C * N <= 5 * 16 * 8 * 192 * 256:
        rng = np.random.RandomState(NB + C + H)
        x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
        rng.normal(size=(3, 3, C, N))
        dy = rng.normal(size=(NB, H, W, N)).astype(np.float32)
        xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)), mode='reflect' if pad else 'constant')
        g = np.stack([np.stack([np.einsum('bhwc,bhwn->cn', xp[:, i:i + H, j:j + W], dy.astype(np.float64)) for j in range(3)]) for i in range(3)])
        ref = ' vs f64 %.2e' % (np.max(np.abs(a - g)) / np.max(np.abs(g)))
    bad += err > (5e-6 if os.environ.get('DTYPE', 'f32x3') == 'f32x3' else 2e-5)
    print('%-24s err %.2e%s  %s | %s' % (key, err, ref, dirs['rowhalo'][1][key], dirs['pertap'][1][key]))
print('FAIL' if bad else 'OK')
sys.exit(1 if bad else 0)
