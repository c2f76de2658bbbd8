"""Per-shape GEMM kernel throughput (HIP events around the launches, via the profile API)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea

SHAPES = [  # name, NB,H,W,C,N,k,stride,pad
    ('D.conv1 patch fwd 4x4s2 64->128 @128^2/2', 128, 64, 64, 64, 128, 4, 2, 0),
    ('D.conv3 patch 4x4s2 128->256', 128, 32, 32, 128, 256, 4, 2, 0),
    ('D.conv4 patch 4x4s2 256->512', 128, 16, 16, 256, 512, 4, 2, 0),
    ('D.conv4 joint 4x4s2 256->512', 128, 8, 16, 256, 512, 4, 2, 0),
    ('E.res 3x3 REFLECT 128->128 @32', 64, 32, 32, 128, 128, 3, 1, 1),
    ('E.res 3x3 REFLECT 512->512 @4', 64, 4, 4, 512, 512, 3, 1, 1),
    ('cls Conv_2 u1 3x3 128->128 @64', 64, 64, 64, 128, 128, 3, 1, 0),
    ('cls Conv_2 u2 3x3 256->256 @32', 64, 32, 32, 256, 256, 3, 1, 0),
    ('cls Conv_2 u3 3x3 512->512 @16', 64, 16, 16, 512, 512, 3, 1, 0),
    ('cls Conv_2 u4 3x3 768->768 @8', 64, 8, 8, 768, 768, 3, 1, 0),
    ('cls Conv_1 u1 3x3 8->128 @64', 64, 64, 64, 8, 128, 3, 1, 0),
    ('cls h0 7x7 3->8 @64', 64, 64, 64, 3, 8, 7, 1, 0),
    ('D.conv0 patch 4x4s2 3->64', 128, 128, 128, 3, 64, 4, 2, 0),
]


def main():
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype=os.environ.get('DTYPE', 'f32')))
    rng = np.random.RandomState(0)
    print(f'{"shape":44s} {"kernel":26s} {"ms":>8s} {"TF/s":>8s}')
    only = os.environ.get('SHAPES')
    for name, NB, H, W, C, N, k, s, pad in SHAPES:
        if only and not any(t in name for t in only.split(',')):
            continue
        x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
        w = (rng.normal(size=(k, k, C, N)) * 0.05).astype(np.float32)
        if os.environ.get('OPERANDS') == 'zero':       # power probe: the same launches on all-zero operands (no toggling in the
            x[...] = 0                                 # matrix pipe's data paths): a kernel that speeds up is power-limited
            w[...] = 0
        elif os.environ.get('OPERANDS') == 'const':
            x[...] = 1.5
            w[...] = 0.75
        OH, OW = (H + s - 1) // s, (W + s - 1) // s
        dy = rng.normal(size=(NB, OH, OW, N)).astype(np.float32)
        shapes = [(NB, OH, OW, N), x.shape, w.shape, None]
        net.test_op('conv2d', [x, w, None, dy], [NB, H, W, C, N, k, s, pad], shapes)   # warm-up
        net.profile_begin()
        for _ in range(3):
            net.test_op('conv2d', [x, w, None, dy], [NB, H, W, C, N, k, s, pad], shapes)
        prof = net.profile_end()
        for kq in sorted(prof['kernels'], key=lambda q: -q['ms']):
            print(f'{name:44s} {kq["name"]:26s} {kq["ms"] / kq["launches"]:8.3f} {kq["flops"] / kq["ms"] / 1e9:8.1f}')
    net.close()


if __name__ == '__main__':
    main()
