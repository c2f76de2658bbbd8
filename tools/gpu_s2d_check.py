"""S2D (space-to-depth halo form of the 4x4 stride-2 forward convolution) against the tap-gather kernel, per shape and
mode: both kernels evaluate the same limb / bf16 products, so the outputs must agree to fp32 summation order."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(8, 64, 64, 64, 128), (64, 32, 32, 128, 256), (4, 32, 32, 128, 256), (64, 16, 16, 256, 512),
          (16, 32, 32, 128, 128), (16, 32, 32, 64, 256), (256, 32, 32, 128, 128), (512, 32, 32, 128, 128), (128, 32, 32, 128, 128),
          (32, 32, 32, 128, 128), (64, 32, 32, 96 + 32, 128), (64, 32, 32, 32, 128), (64, 64, 64, 64, 128)]
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import numpy as np
    import edgegan_amd as ea
    mode = sys.argv[2]
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype=mode))
    out = {}
    for (NB, H, W, C, N) in SHAPES:
        rng = np.random.RandomState(NB + C)
        x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
        w = (rng.normal(size=(4, 4, C, N)) * 0.05).astype(np.float32)
        net.profile_begin()
        y, = net.test_op('conv2d', [x, w, None, None], [NB, H, W, C, N, 4, 2, 0], [(NB, H // 2, W // 2, N)])[:1]
        lay = net.profile_layers().strip().splitlines()[-1]
        k = net.profile_end()['kernels']
        np.save(os.path.join(sys.argv[3], f'{mode}_{NB}_{H}_{C}_{N}.npy'), y)
        out[f'{NB},{H},{W},{C},{N}'] = [q['name'] for q in k] + ['split=' + lay.split('",')[1].split(',')[5]]
    print('KERNELS ' + json.dumps(out))
    net.close()
    sys.exit(0)
import numpy as np, tempfile
for mode in ('f32x3', 'bf16'):
    dirs = {}
    for tag, env in (('s2d', {}), ('ref', {'EG_NO_S2D': '1'})):
        d = tempfile.mkdtemp()
        e = dict(os.environ, EG_LIB_PATH=os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so'), **env)
        r = subprocess.run([sys.executable, __file__, 'child', mode, d], env=e, capture_output=True, text=True)
        if r.returncode != 0:
            print(mode, tag, 'FAILED', r.stderr[-2000:])
            sys.exit(1)
        dirs[tag] = (d, json.loads([l for l in r.stdout.splitlines() if l.startswith('KERNELS ')][-1][8:]))
    for (NB, H, W, C, N) in SHAPES:
        a = np.load(os.path.join(dirs['s2d'][0], f'{mode}_{NB}_{H}_{C}_{N}.npy'))
        b = np.load(os.path.join(dirs['ref'][0], f'{mode}_{NB}_{H}_{C}_{N}.npy'))
        key = f'{NB},{H},{W},{C},{N}'
        err = float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if np.all(np.isfinite(a)) else float('inf')
        print('%-6s %-22s err %.2e  %s | %s' % (mode, key, err, dirs['s2d'][1][key], dirs['ref'][1][key]))
