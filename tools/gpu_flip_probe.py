"""Which near-kink elements explain a full-batch deviation?  For one sub-step at B = 64: the engine's filter gradients against the
fp64 oracle per OUTPUT CHANNEL, the oracle's small-extent activation inputs closest to the kink in the worst channels, and the
error after flipping each of them (oracle/edgegan_oracle.py: kink_flipper).   python tools/gpu_flip_probe.py d_optim_patch2 D_patch2/d_conv_4/conv2d/w"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
torch.set_num_threads(min(16, os.cpu_count() or 1))
import edgegan_amd as ea, parity, full_batch_lib as FB, test_tight_gpu as TT
O = parity.O
name, var = sys.argv[1], sys.argv[2]
margin = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
case = FB.CASE['b64/' + name]
cfg = ea.Config(**dict(case[2], compute_dtype=os.environ.get('EG_MODE', 'f32x3')))
values = ea.init_variables(cfg, case[3]); b = ea.synthetic_batch(cfg, seed=case[4])
net = ea.EdgeGAN(cfg, values); net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
net.substep(TT.SUBSTEP_INDEX[name], apply=False)
h = net.get_variable(var + '/grad').astype(np.float64)
om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
in_calls = []     # EG_IN_PROBE: per InstanceNorm call, its sigma / rms per (n, c)
if os.environ.get('EG_IN_PROBE'):
    _in = O.instance_norm
    def _probe(x):
        d = x.detach()
        sig = d.var(dim=(1, 2), unbiased=False).sqrt()          # [B][C]
        in_calls.append((tuple(d.shape), sig / d.pow(2).mean().sqrt()))
        return _in(x)
    O.instance_norm = _probe
with O.kink_recorder(collect=(TT.SF.SMALL_EXTENT, margin)) as rec:
    g64, _ = om.compute_grads(name, inp, inp.eps[0])
if in_calls:
    O.instance_norm = _in
g = g64[var].numpy()
print('rel L2', parity.rel_l2(h, g), 'shape', g.shape, 'near-kink candidates inside', margin, ':', len(rec.near))
e = (h - g).reshape(-1, g.shape[-1]); per_n = np.sqrt((e ** 2).sum(0)); tot = np.sqrt((e ** 2).sum())
top = np.argsort(-per_n)[:6]
print('error share of the worst output channels:', [(int(n), round(float(per_n[n] / tot), 3)) for n in top])
ec = (h - g).reshape(-1, g.shape[-2], g.shape[-1]); per_c = np.sqrt((ec ** 2).sum((0, 2)))
topc = np.argsort(-per_c)[:6]
print('error share of the worst input channels:', [(int(c), round(float(per_c[c] / tot), 3)) for c in topc])
for k, (shp, rel) in enumerate(in_calls):     # the worst output channel's InstanceNorm conditioning, where the channel count matches
    if shp[-1] == g.shape[-1] and shp[1] * shp[2] <= 256:
        col = rel[:, int(top[0])]
        print(f'instance_norm call {k} {shp}: sigma/rms of channel {int(top[0])}: min {float(col.min()):.2e} (image {int(col.argmin())}), median {float(col.median()):.2e};'
              f' over all channels min {float(rel.min()):.2e}')
# calls by tag / shape
calls = {}
for c, i, m, tag, shp in rec.near:
    calls.setdefault(c, []).append((i, m, tag, shp))
print('calls with candidates:', {c: (len(v), v[0][2], v[0][3]) for c, v in calls.items()})
cand = sorted(rec.near, key=lambda r: r[2])
if os.environ.get('EG_FLIP_CHANNELS'):    # only the candidates in the worst output channels (flat NHWC index % N)
    N = g.shape[-1]
    cand = [r for r in cand if r[4][-1] == N and int(r[1] % N) in [int(n) for n in top[:1]] and r[0] >= int(os.environ.get('EG_FLIP_MIN_CALL', '0'))]
    print('candidates in channels', [int(n) for n in top[:2]], ':', [(c, i, '%.1e' % m, shp) for c, i, m, _, shp in cand])
shown = 0
for c, i, m, tag, shp in cand:
    with O.kink_flipper([(c, i)]):
        gf, _ = om.compute_grads(name, inp, inp.eps[0])
    d = gf[var].numpy() - g
    if np.abs(d).max() == 0 and not os.environ.get('EG_FLIP_ZERO'):
        continue
    err = parity.rel_l2(h, g + d)
    print(f'flip call {c} {shp} element {i} ({tag}, |x|/rms {m:.1e}): delta rel L2 {np.sqrt((d**2).sum())/np.sqrt((g**2).sum()):.2e} -> engine vs flipped oracle {err:.3e}')
    shown += 1
    if shown >= int(os.environ.get('EG_FLIP_SHOW', '12')):
        break
