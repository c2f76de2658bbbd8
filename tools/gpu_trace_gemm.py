"""One training iteration at the bench configuration with EG_TRACE_GEMM=1: aggregates the GEMM launches by shape."""
import os, sys, subprocess, collections, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get('EG_TRACE_GEMM') is None:
    env = dict(os.environ, EG_TRACE_GEMM='1')
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    agg = collections.OrderedDict()
    for line in out.stderr.splitlines():
        if line.startswith('EG_GEMM'):
            key = re.sub(r' gflop=\S+', '', line[8:])
            g = float(line.rsplit('gflop=', 1)[1])
            c = agg.setdefault(key, [0, 0.0]); c[0] += 1; c[1] += g
    tot = sum(v[1] for v in agg.values())
    for k, (n, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:4d}x {g:9.2f} GF {100 * g / tot:5.1f}%  {k}')
    print('total GFLOP', tot)
    sys.exit(0)
sys.path.insert(0, ROOT)
import edgegan_amd as ea
cfg = ea.Config(batch_size=int(os.environ.get('B', '64')), compute_dtype=os.environ.get('DTYPE', 'f32'))
net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
b = ea.synthetic_batch(cfg)
net.upload_batch(**b)
net.train_step(); net.sync()
net.close()
