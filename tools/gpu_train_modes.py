#!/usr/bin/env python
"""The same 300 training iterations (batch 16, 14 classes, a fresh synthetic batch every iteration, same seeds) in the three arithmetic
modes: the twelve losses at checkpoints side by side, and how far the modes' weights have drifted apart.  Evidence that the modes
train alike (they are different roundings of one computation on a chaotic trajectory: agreement is statistical, not bitwise).
usage (GPU box): python tools/gpu_train_modes.py [iterations] > out.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    marks = sorted({1, 2, 5, 10, 20, 50, 100, 200, iters} & set(range(1, iters + 1)))
    hist, final = {}, {}
    for dtype in ('f32', 'f32x3', 'bf16'):
        cfg = ea.Config(batch_size=16, compute_dtype=dtype)
        net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
        rows = {}
        for it in range(1, iters + 1):
            net.upload_batch(**ea.synthetic_batch(cfg, seed=1000 + it))
            net.train_step()
            if it in marks:
                l = net.losses()
                assert all(np.isfinite(v) for v in l.values()), (dtype, it, l)
                rows[it] = l
        hist[dtype] = rows
        final[dtype] = net.get_variables()
        net.close()
    keys = list(next(iter(hist['f32'].values())).keys())
    for it in marks:
        print(f'iteration {it}')
        for k in keys:
            print('   %-18s' % k + ''.join('  %s %12.5f' % (d, hist[d][it][k]) for d in hist))
    print('relative L2 distance of the trained weights after %d iterations (per network group, against exact fp32):' % iters)
    groups = sorted({n.split('/')[0] for n in final['f32']})
    for g in groups:
        out = []
        for d in ('f32x3', 'bf16'):
            num = sum(float(((final[d][n].astype(np.float64) - final['f32'][n]) ** 2).sum()) for n in final['f32'] if n.split('/')[0] == g)
            den = sum(float((final['f32'][n].astype(np.float64) ** 2).sum()) for n in final['f32'] if n.split('/')[0] == g)
            out.append('%s %.3e' % (d, (num / den) ** 0.5))
        print('   %-10s' % g + '   '.join(out))


if __name__ == '__main__':
    main()
