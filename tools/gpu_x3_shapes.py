"""Conv op parity of the classifier's / discriminators' real layer shapes at small batch in f32x3 mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import edgegan_amd as ea
import parity

net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='f32x3'))
cases = []
for nb in (2, 3):
    cases += [(f'u3 Conv_2 512->512 @16 nb{nb}', nb, 16, 16, 512, 512, 3, 1, 0, True),
              (f'u3 Conv_1 256->512 @16 nb{nb}', nb, 16, 16, 256, 512, 3, 1, 0, True),
              (f'u3 Conv_3 1x1 256->512 @16 nb{nb}', nb, 16, 16, 256, 512, 1, 1, 0, True),
              (f'u3 gate 256->256 @16 nb{nb}', nb, 16, 16, 256, 256, 3, 1, 0, True),
              (f'u2 Conv_2 256->256 @32 nb{nb}', nb, 32, 32, 256, 256, 3, 1, 0, True),
              (f'u4 Conv_2 768->768 @8 nb{nb}', nb, 8, 8, 768, 768, 3, 1, 0, True),
              (f'u1 Conv_2 128->128 @64 nb{nb}', nb, 64, 64, 128, 128, 3, 1, 0, True)]
for c in cases:
    for r in parity.check_conv(net, c, seed=5):
        print(f'{r[1]:10.2e} tol {r[2]:6.0e} {"FAIL" if r[1] > r[2] else "    "} {r[0]}')
