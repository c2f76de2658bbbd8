"""What would stacking the discriminator passes of one sub-step buy?  Per layer: forward + input gradient + filter gradient at
NB = 64 (x-hat / tangent), 128 (real|fake) and 256 (all four stacked), HIP-event times through the profile API."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea

LAYERS = [('Dp conv1 64->128 @64', 64, 64, 64, 128), ('Dp conv3 128->256 @32', 32, 32, 128, 256), ('Dp conv4 256->512 @16', 16, 16, 256, 512),
          ('D conv1 64->128 @32x64', 32, 64, 64, 128), ('D conv3 128->256 @16x32', 16, 32, 128, 256), ('D conv4 256->512 @8x16', 8, 16, 256, 512)]
net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype=os.environ.get('DTYPE', 'f32x3')))
rng = np.random.RandomState(0)
for name, H, W, C, N in LAYERS:
    res = {}
    for NB in (64, 128, 256):
        x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
        w = (rng.normal(size=(4, 4, C, N)) * 0.05).astype(np.float32)
        dy = rng.normal(size=(NB, H // 2, W // 2, N)).astype(np.float32)
        shapes = [(NB, H // 2, W // 2, N), x.shape, w.shape, None]
        net.test_op('conv2d', [x, w, None, dy], [NB, H, W, C, N, 4, 2, 0], shapes)
        net.profile_begin()
        for _ in range(3):
            net.test_op('conv2d', [x, w, None, dy], [NB, H, W, C, N, 4, 2, 0], shapes)
        prof = net.profile_end()
        res[NB] = sum(k['ms'] / 3 for k in prof['kernels'])
        ks = ' '.join(f"{k['name']}:{k['ms'] / k['launches']:.3f}" for k in prof['kernels'])
        print(f'{name:28s} NB={NB:3d} total {res[NB]:.3f} ms  [{ks}]')
    # a sub-step today: fwd at 64+128+64, dgrad at 64 (first order) + 64+128+64, wgrad at 64+128+64
    print(f'{name:28s} separate (2 x 64 + 128) = {2 * res[64] + res[128]:.3f} ms   stacked 256 = {res[256]:.3f} ms')
net.close()
