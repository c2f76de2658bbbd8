"""Golden vectors of the EdgeGAN hot path (SURVEY.md 8c items i-v) -- TEST INFRASTRUCTURE.

The reference ships no tests and no golden vectors and cannot run here (tensorflow_gpu==1.14.0), so these
files do not pin the oracle to the reference ("parity unpinned" stays).  What they do: freeze the fp64 outputs
of oracle/edgegan_oracle.py at one commit, so that
  * a later edit of the oracle that changes any number is caught (tests/test_golden_cpu.py recomputes every
    entry and compares at 1e-11 -- bit-exact on the generating host; the slack only covers a different
    summation order inside torch's threaded fp64 kernels on another host);
  * the HIP path is compared with COMMITTED numbers, not with whatever the oracle computes at test time
    (tests/test_golden_gpu.py).
tests/golden/make_goldens.py writes the files; this module holds the case definitions and the compute
functions shared by the generator and the two tests.  Inputs are regenerated from seeds (numpy RandomState is
stable across versions), outputs are stored: small tensors whole, large ones as (sum, L2 norm, strided slice).
"""
from __future__ import annotations

import os

import numpy as np
import torch

import edgegan_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DT = torch.float64


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=DT, requires_grad=grad)


def digest(a, k: int = 32) -> np.ndarray:
    """[sum, L2 norm, numel, k strided samples] of an array (fp64)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, min(a.size, k)).astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), a.size], a[idx]])


def digest_err(got, want_digest, floor: float = 0.0):
    """Compares an array with a stored digest: returns (relative error of the samples in max-norm scaled by the
    largest sample, relative error of the L2 norm)."""
    d = digest(got, len(want_digest) - 3)
    w = np.asarray(want_digest, dtype=np.float64)
    if d.shape != w.shape or d[2] != w[2] or not np.all(np.isfinite(d)):
        return float('inf'), float('inf')
    scale = max(float(np.max(np.abs(w[3:]))), floor, 1e-300)
    return float(np.max(np.abs(d[3:] - w[3:])) / scale), float(abs(d[1] - w[1]) / max(w[1], floor, 1e-300))


# ------------------------------------------------------------------------------------------------
# (i) per-op forward/backward on tiny shapes (odd sizes, borders) + (v) known-answer vectors
# ------------------------------------------------------------------------------------------------
# name, NB, H, W, C, N, k, stride, pad (0 SAME, 1 REFLECT), bias, with_grad
CONV_CASES = [
    ('same_k4s2', 2, 6, 8, 3, 5, 4, 2, 0, False, True),
    ('same_k4s2_odd_fwd', 1, 7, 9, 3, 4, 4, 2, 0, False, False),     # odd extents: ceil(n/2) outputs, forward only
    ('same_k3s1_odd', 2, 5, 7, 4, 6, 3, 1, 0, True, True),
    ('same_k7s1', 1, 6, 6, 3, 8, 7, 1, 0, True, True),
    ('same_k1s1', 2, 3, 5, 8, 4, 1, 1, 0, True, True),
    ('reflect_k3', 2, 5, 6, 4, 4, 3, 1, 1, True, True),
    ('reflect_k3_2x2', 3, 2, 2, 16, 16, 3, 1, 1, True, True),          # encoder top block: mirror spans the tensor
    ('reflect_k1', 1, 4, 4, 8, 8, 1, 1, 1, True, True),
    ('linear_114', 3, 1, 1, 114, 20, 1, 1, 0, True, True),
]
# name, NB, H, W, Cin, Cout
DECONV_CASES = [('deconv_3x4', 2, 3, 4, 6, 4), ('deconv_to3', 1, 4, 4, 16, 3), ('deconv_1x1', 2, 1, 1, 8, 8)]
ACTS = {0: None, 1: 'relu', 2: 'lrelu', 3: 'tanh'}


def _rng(name: str) -> np.random.RandomState:
    import zlib
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)


def conv_case(case):
    name, NB, H, W, C, N, k, s, pad, bias, wg = case
    rng = _rng('conv/' + name)
    x = rng.normal(size=(NB, H, W, C)).astype(np.float32)
    w = (rng.normal(size=(k, k, C, N)) * 0.1).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32) if bias else None
    xt, wt = t64(x, True), t64(w, True)
    bt = t64(b, True) if bias else None
    y = O.conv2d(xt, wt, s, 'REFLECT' if pad else 'SAME', bt)
    out = {'y': y.detach().numpy()}
    dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
    if wg:
        gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, wt] + ([bt] if bias else []))
        out.update(dx=gr[0].numpy(), dw=gr[1].numpy())
        if bias:
            out['db'] = gr[2].numpy()
    return {'x': x, 'w': w, 'b': b, 'dy': dy if wg else None}, out


def deconv_case(case):
    name, NB, H, W, Cin, Cout = case
    rng = _rng('deconv/' + name)
    x = rng.normal(size=(NB, H, W, Cin)).astype(np.float32)
    w = (rng.normal(size=(5, 5, Cout, Cin)) * 0.1).astype(np.float32)
    b = rng.normal(size=(Cout,)).astype(np.float32)
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    y = O.deconv2d(xt, wt, bt, (2 * H, 2 * W))
    dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
    gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, wt, bt])
    return {'x': x, 'w': w, 'b': b, 'dy': dy}, {'y': y.detach().numpy(), 'dx': gr[0].numpy(), 'dw': gr[1].numpy(),
                                                 'db': gr[2].numpy()}


def _unary(name, shape, fn, scale=1.0, shift=0.0):
    rng = _rng(name)
    x = (rng.normal(size=shape) * scale + shift).astype(np.float32)
    xt = t64(x, True)
    y = fn(xt)
    dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
    gx, = torch.autograd.grad((y * t64(dy)).sum(), xt)
    return {'x': x, 'dy': dy}, {'y': y.detach().numpy(), 'dx': gx.numpy()}


def minmax_ref(t):
    mn, mx = t.amin(dim=(1, 2), keepdim=True), t.amax(dim=(1, 2), keepdim=True)
    return (t - mn) / (mx - mn)


# engine op name, shape, extra int param, reference function
UNARY_CASES = {
    'in_act_none': ('in_act', (2, 3, 5, 8), 0, lambda t: O.instance_norm(t), 1.7, 0.3),
    'in_act_relu': ('in_act', (2, 4, 4, 16), 1, lambda t: O.activation(O.instance_norm(t), 'relu'), 1.7, 0.3),
    'in_act_lrelu': ('in_act', (3, 2, 2, 32), 2, lambda t: O.activation(O.instance_norm(t), 'lrelu'), 1.7, 0.3),
    'act_tanh': ('act', (2, 3, 3, 3), 3, lambda t: torch.tanh(t), 1.0, 0.0),
    'act_lrelu': ('act', (2, 3, 3, 8), 2, lambda t: O.activation(t, 'lrelu'), 1.0, 0.0),
    'act_relu': ('act', (2, 3, 3, 8), 1, lambda t: O.activation(t, 'relu'), 1.0, 0.0),
    'resize_4_to_8': ('resize', (2, 4, 4, 3), 8, lambda t: O.resize_bicubic(t, 8), 1.0, 0.0),
    'resize_6_to_12': ('resize', (1, 6, 6, 3), 12, lambda t: O.resize_bicubic(t, 12), 1.0, 0.0),
    'resize_8_to_4': ('resize', (2, 8, 8, 3), 4, lambda t: O.resize_bicubic(t, 4), 1.0, 0.0),
    'pool2': ('pool2', (2, 4, 6, 5), 0, lambda t: O.avg_pool2(t), 1.0, 0.0),
    'gmean_relu_2x2': ('gmean', (3, 2, 2, 8), 1,
                       lambda t: O.avg_pool_same(O.relu(t), 8).reshape(t.shape[0], 1, 1, t.shape[3]), 1.0, 0.0),
    'gmean_4x4': ('gmean', (2, 4, 4, 6), 0, lambda t: t.mean(dim=(1, 2), keepdim=True), 1.0, 0.0),
    'minmax': ('minmax', (2, 4, 4, 8), 0, minmax_ref, 1.0, 0.0),
}


def unary_case(key):
    op, shape, extra, fn, sc, sh = UNARY_CASES[key]
    return _unary('unary/' + key, shape, fn, sc, sh)


def in_tangent_case():
    rng = _rng('in_tangent')
    shape = (2, 4, 4, 8)
    x = (rng.normal(size=shape) * 1.3).astype(np.float32)
    tx = rng.normal(size=shape).astype(np.float32)
    dty = rng.normal(size=shape).astype(np.float32)
    xt, txt = t64(x, True), t64(tx, True)
    f = lambda u: O.activation(O.instance_norm(u), 'lrelu')
    y, ty = torch.autograd.functional.jvp(f, (xt,), (txt,), create_graph=True)
    gx, gtx = torch.autograd.grad((ty * t64(dty)).sum(), [xt, txt])
    return {'x': x, 'tx': tx, 'dty': dty}, {'y': y.detach().numpy(), 'ty': ty.detach().numpy(), 'dtx': gtx.numpy(),
                                            'dx': gx.numpy()}


def bn_case():
    rng = _rng('bn')
    shape = (3, 2, 2, 8)
    x = rng.normal(size=shape).astype(np.float32)
    g = rng.uniform(0.5, 1.5, size=(8,)).astype(np.float32)
    b = (rng.normal(size=(8,)) * 0.3).astype(np.float32)
    dy = rng.normal(size=shape).astype(np.float32)
    xt, gt, bt = t64(x, True), t64(g, True), t64(b, True)
    yb, mean, var = O.batch_norm_train(xt, gt, bt)
    y = O.relu(yb)
    gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, gt, bt])
    return {'x': x, 'gamma': g, 'beta': b, 'dy': dy}, {'y': y.detach().numpy(), 'dx': gr[0].numpy(),
                                                      'dgamma': gr[1].numpy(), 'dbeta': gr[2].numpy(),
                                                      'mean': mean.detach().numpy(), 'var': var.detach().numpy()}


def prelu_case():
    rng = _rng('prelu')
    x = rng.normal(size=(300,)).astype(np.float32)
    lk = np.array([0.2], dtype=np.float32)
    dy = rng.normal(size=(300,)).astype(np.float32)
    xt, lt = t64(x, True), t64(lk[0], True)
    y = O.prelu(xt, lt)
    gx, gl = torch.autograd.grad((y * t64(dy)).sum(), [xt, lt])
    return {'x': x, 'leak': lk, 'dy': dy}, {'y': y.detach().numpy(), 'dx': gx.numpy(), 'dleak': gl.numpy().reshape(1)}


def linear1_case():
    rng = _rng('linear1')
    x = rng.normal(size=(3, 96)).astype(np.float32)
    w = (rng.normal(size=(96, 1)) * 0.05).astype(np.float32)
    b = np.array([0.3], dtype=np.float32)
    dy = rng.normal(size=(3, 1)).astype(np.float32)
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    y = O.linear(xt, wt, bt)
    gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, wt, bt])
    return {'x': x, 'w': w, 'b': b, 'dy': dy}, {'y': y.detach().numpy(), 'dx': gr[0].numpy(), 'dw': gr[1].numpy(),
                                                 'db': gr[2].numpy()}


def sn_case(K=27, Co=8):
    rng = _rng(f'sn/{K}x{Co}')
    W = (rng.normal(size=(K, Co)) * 0.02).astype(np.float32)
    u = np.clip(rng.normal(size=(1, Co)), -2, 2).astype(np.float32)
    dWb = rng.normal(size=(K, Co)).astype(np.float32)
    Wt = t64(W, True)
    Wbar = O.spectral_normed_weight(Wt, t64(u))
    gW, = torch.autograd.grad((Wbar * t64(dWb)).sum(), Wt)
    sigma = float((W.astype(np.float64) / Wbar.detach().numpy())[0, 0])
    return {'W': W, 'u': u, 'dWbar': dWb}, {'Wbar': Wbar.detach().numpy(), 'dW': gW.numpy(), 'sigma': np.array([sigma])}


def acgan_case(mode):
    rng = _rng('acgan')
    B, NC = 6, 14
    lg = (rng.normal(size=(B, NC)) * 2).astype(np.float32)
    lab = rng.randint(0, NC, size=(B,))
    lt = t64(lg, True)
    lg_, ld_ = O.acgan_loss_focal(lt, torch.tensor(lab), lt)
    ref = ld_ if mode == 0 else lg_
    gl, = torch.autograd.grad(ref, lt)
    return {'logits': lg, 'labels': lab.astype(np.float32)}, {'loss': ref.detach().numpy().reshape(1), 'dlogits': gl.numpy()}


def gp_case(lambda_gp=10.0):
    rng = _rng('gp')
    B, per = 3, 50
    g = (rng.normal(size=(B, per)) * 0.2).astype(np.float32)
    gt = t64(g, True)
    nr = torch.sqrt((gt ** 2).sum(dim=1))
    gp = lambda_gp * torch.mean((nr - 1) ** 2)
    v, = torch.autograd.grad(gp, gt)
    return {'g': g}, {'v': v.numpy(), 'norms': nr.detach().numpy(), 'gp': gp.detach().numpy().reshape(1)}


def rmsprop_case():
    rng = _rng('rmsprop')
    n = 64
    w, g = rng.normal(size=(n,)).astype(np.float32), rng.normal(size=(n,)).astype(np.float32)
    ms = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    mom = np.zeros(n, dtype=np.float32)
    wt, gt, mst, momt = t64(w), t64(g), t64(ms), t64(mom)
    O.rmsprop_apply(wt, gt, mst, momt, 2e-4)
    return {'w': w, 'g': g, 'ms': ms, 'mom': mom}, {'w': wt.numpy(), 'ms': mst.numpy(), 'mom': momt.numpy()}


def l1_case():
    rng = _rng('l1')
    z = rng.normal(size=(3, 100)).astype(np.float32)
    r = rng.normal(size=(3, 100)).astype(np.float32)
    rt = t64(r, True)
    loss = O.l1loss(t64(z), rt, 10.0)
    g, = torch.autograd.grad(loss, rt)
    return {'z': z, 'recon': r}, {'loss': loss.detach().numpy().reshape(1), 'drecon': g.numpy()}


def known_answers():
    """(v): bicubic and spectral-norm known-answer vectors + TF SAME pads."""
    R = O.bicubic_matrix(64, 128)
    W = np.arange(12, dtype=np.float64).reshape(4, 3) / 10.0 - 0.4
    u = np.array([[1.0, -0.5, 0.25]])
    Wbar = O.spectral_normed_weight(t64(W), t64(u)).numpy()
    return {'bicubic_64_128_rows_10_11_127': np.stack([R[10], R[11], R[127]]),
            'bicubic_8_4': O.bicubic_matrix(8, 4),
            'sn_4x3_Wbar': Wbar,
            'same_pads': np.array([O.same_pads(64, 4, 2), O.same_pads(64, 3, 1), O.same_pads(64, 7, 1),
                                   O.same_pads(64, 5, 2), O.same_pads(2, 8, 8), O.same_pads(7, 4, 2)], dtype=np.float64)}


def compute_ops():
    """{key: fp64 array} for every op golden.  Keys: '<case>/<output>'."""
    out = {}

    def put(prefix, res):
        for k, v in res[1].items():
            out[f'{prefix}/{k}'] = np.asarray(v, dtype=np.float64)

    for c in CONV_CASES:
        put('conv2d/' + c[0], conv_case(c))
    for c in DECONV_CASES:
        put('deconv2d/' + c[0], deconv_case(c))
    for k in UNARY_CASES:
        put('unary/' + k, unary_case(k))
    put('in_tangent', in_tangent_case())
    put('bn_relu', bn_case())
    put('prelu', prelu_case())
    put('linear1', linear1_case())
    put('sn/27x8', sn_case(27, 8))
    put('sn/288x32', sn_case(288, 32))
    put('acgan/mode0', acgan_case(0))
    put('acgan/mode1', acgan_case(1))
    put('gp', gp_case())
    put('rmsprop', rmsprop_case())
    put('l1', l1_case())
    for k, v in known_answers().items():
        out['known/' + k] = v
    return out


# ------------------------------------------------------------------------------------------------
# (ii) each network forward at B=2 (full-size architectures), digests
# ------------------------------------------------------------------------------------------------
def net_inputs(cfg, seed=20):
    """Same draws as tests/parity.py::_net_checks."""
    rng = np.random.RandomState(seed)
    NB = cfg.batch_size
    g_in = cfg.z_dim + (cfg.num_classes if cfg.multiclasses else 0)
    zin = rng.normal(size=(NB, g_in)).astype(np.float32)
    x = rng.uniform(-1, 1, size=(NB, cfg.height, cfg.width, 3)).astype(np.float32)
    xp = rng.uniform(-1, 1, size=(NB, cfg.dis_size, cfg.dis_size, 3)).astype(np.float32)
    xe = rng.uniform(-1, 1, size=(NB, cfg.height, cfg.width // 2, 3)).astype(np.float32)
    return {'zin': zin, 'x': x, 'xp': xp, 'xe': xe}


def compute_nets(ocfg, values, inputs, eps=0.37):
    p = {k: t64(v) for k, v in values.items()}
    out = {}
    for g in ('G1', 'G2'):
        out[g] = digest(O.generator(p, g, t64(inputs['zin']), ocfg).numpy(), 256)
    out['D'] = O.discriminator(p, 'D', t64(inputs['x']))[1].numpy().astype(np.float64)
    for d in ('D_patch2', 'D_patch3'):
        out[d] = O.discriminator(p, d, t64(inputs['xp']))[1].numpy().astype(np.float64)
    out['E'] = O.encoder(p, t64(inputs['xe']), eps, ocfg)[0].numpy().astype(np.float64)
    if ocfg.multiclasses:
        out['D2'] = O.classifier(p, t64(inputs['xe']), ocfg).numpy().astype(np.float64)
    return out


# ------------------------------------------------------------------------------------------------
# (iii) the 7 sub-steps, free-running for `iterations` iterations in fp64
# ------------------------------------------------------------------------------------------------
LOSS_KEYS = {'d_optim': ['joint_dis_dloss'], 'd_optim_patch2': ['image_dis_dloss'], 'd_optim_patch3': ['edge_dis_dloss'],
             'd_optim2': ['loss_d_ac'], 'g_optim_u': ['edge_gloss', 'image_gloss'], 'e_optim': ['zl_loss'],
             'g_optim_b': ['edge_gloss', 'image_gloss']}


def compute_steps(ocfg, values, batch, iterations=2):
    """Oracle free run: per (iteration, sub-step): losses, and per optimised variable the digest of its gradient;
    after each iteration the digest of (w - w_initial) for every trainable variable and of the BN moving stats."""
    om = O.OracleModel(ocfg, values, dtype=DT)
    inp = O.StepInputs(t64(batch['images']), t64(batch['z']), t64(batch['alpha']), t64(batch['eps']))
    init = {n: t.clone() for n, t in om.p.items()}
    out = {}
    for it in range(iterations):
        for name in om.substep_names():
            grads, losses = om.compute_grads(name, inp, inp.eps[0])
            for lk in LOSS_KEYS[name]:
                out[f'it{it}/{name}/loss/{lk}'] = np.array([float(losses[lk])])
            for vn, g in grads.items():
                out[f'it{it}/{name}/grad/{vn}'] = digest(g.numpy(), 16)
            om.apply_grads(grads)
            om.update_bn_moving(om._GENS[name])
        for vn, t in om.p.items():
            if om.trainable[vn]:
                out[f'it{it}/delta/{vn}'] = digest((t - init[vn]).numpy(), 16)
            elif 'moving_' in vn:
                out[f'it{it}/bn/{vn}'] = digest(t.numpy(), 16)
    return out


# ------------------------------------------------------------------------------------------------
# (iv) inference E -> G1, G2 on the reference's example test PNGs, fixed eps
# ------------------------------------------------------------------------------------------------
def compute_infer(ocfg, values, images_list, classes_list, eps=(0.3, -0.7)):
    om = O.OracleModel(ocfg, values, dtype=DT)
    out = {}
    for i, (imgs, cls) in enumerate(zip(images_list, classes_list)):
        e, im = om.infer(t64(imgs), torch.tensor(cls), eps[0], eps[1])
        out[f'img{i}/edge'] = digest(e.numpy(), 256)
        out[f'img{i}/image'] = digest(im.numpy(), 256)
    return out
