"""Parity checks of the HIP engine against the CPU oracle (oracle/edgegan_oracle.py).

Each check returns rows (name, error, tolerance).  Error metric: max|a-b| / (max|b| + 1e-30) with b the
fp64 oracle value -- a scale-relative max-norm error.  Tolerances (fp32 arithmetic, different
summation order than the oracle): forward 2e-5, backward 5e-5 unless a row says otherwise; the
full-step tolerances are documented at substep_checks below and in tests/test_step_gpu.py / tests/test_tight_gpu.py.
"""
from __future__ import annotations

import numpy as np
import torch

import edgegan_oracle as O
import edgegan_amd as ea

FWD_TOL = 2e-5
BWD_TOL = 5e-5
DT = torch.float64


def relerr(a, b, floor: float = 0.0) -> float:
    """max|a-b| / max(max|b|, floor).  `floor` keeps variables whose true value is (numerically) zero --
    e.g. the gradient of a bias that feeds an InstanceNorm -- from reporting noise/0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float('inf')
    if not np.all(np.isfinite(a)):
        return float('inf')
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), floor, 1e-30))


def rel_l2(a, b, floor: float = 0.0) -> float:
    """||a-b||_2 / max(||b||_2, floor*sqrt(n))."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.shape != b.shape or not np.all(np.isfinite(a)):
        return float('inf')
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), floor * np.sqrt(b.size), 1e-300))


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=DT, requires_grad=grad)


def oracle_config(cfg: ea.Config) -> O.Config:
    return O.Config(batch_size=cfg.batch_size, height=cfg.height, width=cfg.width, num_classes=cfg.num_classes,
                    z_dim=cfg.z_dim, gf_dim=cfg.gf_dim, df_dim=cfg.df_dim, c_dim=cfg.c_dim, dis_size=cfg.dis_size,
                    lambda_gp=cfg.lambda_gp, zl_weight=cfg.zl_weight, learning_rate=cfg.learning_rate,
                    joint_dweight=cfg.joint_dweight, image_dweight=cfg.image_dweight, edge_dweight=cfg.edge_dweight,
                    use_image_discriminator=cfg.use_image_discriminator, use_edge_discriminator=cfg.use_edge_discriminator,
                    image_dis_size=cfg.image_dis_size, edge_dis_size=cfg.edge_dis_size,
                    E_norm=cfg.E_norm, G_norm=cfg.G_norm, D_norm=cfg.D_norm)


# ------------------------------------------------------------------------------------------------
# per-op checks
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name, NB,H,W,C,N,k,stride,pad(0 SAME,1 REFLECT), bias
    ('d_conv_0-like 4x4s2 C3 (flat)', 2, 16, 32, 3, 64, 4, 2, 0, False),
    ('d_conv_1-like 4x4s2 C64 (vec 64x64)', 2, 16, 16, 64, 128, 4, 2, 0, False),
    ('big 3x3s1 C16->128 (vec 128x128)', 4, 128, 128, 16, 128, 3, 1, 0, True),
    ('E res 3x3 REFLECT C64->128 +b', 2, 16, 16, 64, 128, 3, 1, 1, True),
    ('E res 3x3 REFLECT 4x4 C32->32 +b', 2, 4, 4, 32, 32, 3, 1, 1, True),
    ('E shortcut 1x1 REFLECT C64->128 +b', 2, 8, 8, 64, 128, 1, 1, 1, True),
    ('E stem 4x4s2 C3->64 +b', 2, 16, 16, 3, 64, 4, 2, 0, True),
    ('cls h0 7x7 C3->8 (narrow flat)', 2, 16, 16, 3, 8, 7, 1, 0, True),
    ('cls Conv_1 3x3 C8->128 (flat)', 2, 16, 16, 8, 128, 3, 1, 0, True),
    ('cls 3x3 C16->8 (narrow vec)', 2, 16, 16, 16, 8, 3, 1, 0, True),
    ('cls Conv_3 1x1 C128->256', 2, 8, 8, 128, 256, 1, 1, 0, True),
    ('g_lin_0-like linear 114->256', 4, 1, 1, 114, 256, 1, 1, 0, True),
    ('FC8-like linear 512->100', 4, 1, 1, 512, 100, 1, 1, 0, True),
    # batch-row linears on the rows_linear kernel (k_gemm.hip; <= 64 rows, <= 128 contracted, >= 256 columns): two row blocks with
    # a partial one, a partial 64-column block, the smallest contraction
    ('g_lin_0 linear 114->8192 nb 37 (rows_linear)', 37, 1, 1, 114, 8192, 1, 1, 0, True),
    ('FC8 input-gradient-like linear 100->512 nb 64 (rows_linear)', 64, 1, 1, 100, 512, 1, 1, 0, False),
    ('class-head input-gradient-like linear 14->776 nb 5 (rows_linear)', 5, 1, 1, 14, 776, 1, 1, 0, True),
    ('fc-like linear 768->14 (flat N)', 4, 1, 1, 768, 14, 1, 1, 0, True),
    # thin layers at the extents that select the dedicated kernels (k_thin.hip): several strips per image with halo
    # rows, one-row rounds (W = 64), whole-image blocks (8x8), the 3-channel image rows of the MRU units
    ('cls Conv_1 3x3 C8->128 64x64 (strips)', 1, 64, 64, 8, 128, 3, 1, 0, True),
    ('cls mru Conv 3x3 C3->128 32x32 (image rows)', 2, 32, 32, 3, 128, 3, 1, 0, True),
    ('cls mru Conv 3x3 C3->512 8x8', 3, 8, 8, 3, 512, 3, 1, 0, False),
    ('d_conv_0-like 4x4s2 C3->64 64x128', 1, 64, 128, 3, 64, 4, 2, 0, False),
    # ... and the remaining shapes of the packed filter-gradient kernel (k_thin_wgrad.hip): 8 / 3 gathered channels against 8
    # output channels (one 16-column tile, four pixel sub-ranges per block), the 1x1 layer, 32-pixel segments of 64-wide rows
    # under the 7x7 halo, REFLECT padding with 8 four-pixel rows per K step across 3 images
    ('cls gate 3x3 C8->8 32x32 (hidden rows of unit 1)', 2, 32, 32, 8, 8, 3, 1, 0, True),
    ('cls Conv_3 1x1 C8->128 16x16', 2, 16, 16, 8, 128, 1, 1, 0, True),
    ('cls mru Conv 3x3 C3->8 64x64 (32-pixel segments)', 1, 64, 64, 3, 8, 3, 1, 0, True),
    ('cls h0 7x7 C3->8 64x64', 1, 64, 64, 3, 8, 7, 1, 0, True),
    ('thin REFLECT 3x3 C3->32 8x4 nb3', 3, 8, 4, 3, 32, 3, 1, 1, True),
    # conv_scatter in column strips (k_thin.hip: rows wider than its product ring holds -- configs[4]'s 128 / 256-column maps): three
    # strips with a 4-column last one, five strips, a width that is no power of two
    ('thin 3x3 C64->12 32x128 (column strips)', 2, 32, 128, 64, 12, 3, 1, 0, True),
    ('thin 3x3 C128->8 16x256 (column strips)', 1, 16, 256, 128, 8, 3, 1, 0, True),
    ('thin 3x3 C32->4 8x192 (column strips, width 192)', 2, 8, 192, 32, 4, 3, 1, 0, False),
    # stride-1 layers with >= 128 output channels: the halo-staged kernel (k_gemm_x3h.hip) in the f32x3 / bf16 modes --
    # one 16x16 tile per image, 2x2 tiles per image, 4 images per tile, and the 2x2-tap phases of a stride-2 input gradient
    ('cls Conv_2 3x3 C128->128 16x16', 2, 16, 16, 128, 128, 3, 1, 0, True),
    ('cls 3x3 C256->128 32x32 (2x2 tiles)', 1, 32, 32, 256, 128, 3, 1, 0, False),
    ('cls Conv_2 3x3 C64->256 8x8 (4 images per tile)', 4, 8, 8, 64, 256, 3, 1, 0, True),
    ('d_conv_3-like 4x4s2 C128->256 (dgrad phases)', 4, 16, 16, 128, 256, 4, 2, 0, False),
    # ... and its 64-output-channel form (32 x 64 wave tiles): d_conv_1's input gradient at an extent that fills the CUs
    ('d_conv_1-like 4x4s2 C64->128 64x64 (dgrad to 64 channels)', 8, 64, 64, 64, 128, 4, 2, 0, False),
]


def bf16r(t):
    """Round-to-nearest-even to bf16, returned in the oracle dtype (what the engine's bf16 kernels do to operands)."""
    return t.detach().to(torch.float32).to(torch.bfloat16).to(DT)


def bf16_conv_rule(K, n_out):
    """a conv / dgrad / transposed-conv contraction takes bf16 operands iff its contracted channel
    count K is a multiple of 32 and it produces more than 16 channels, a multiple of 4."""
    return K % 32 == 0 and n_out > 16 and n_out % 4 == 0


def bf16_wgrad_rule(C, N):
    """a weight gradient takes bf16 operands iff both channel counts are multiples of 4 and the
    gathered operand has more than 16 channels (the flattened-(tap,c) kernel for C <= 16 stays fp32)."""
    return C % 4 == 0 and N % 4 == 0 and C > 16


def check_conv(net: ea.EdgeGAN, case, seed=0, bf16=False):
    """bf16=True: the engine was created with compute_dtype='bf16'; the oracle then evaluates each of the three
    bilinear maps (forward, input gradient, filter gradient) on bf16-ROUNDED operands in fp64 -- exactly the
    numbers the bf16 MFMA multiplies -- so the tolerances stay the fp32-accumulation ones."""
    name, NB, H, W, Cc, N, k, s, pad, bias = case
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(NB, H, W, Cc)).astype(np.float32)
    w = (rng.normal(size=(k, k, Cc, N)) * 0.1).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32) if bias else None
    xt, wt = t64(x, True), t64(w, True)
    bt = t64(b, True) if bias else None
    mode = 'REFLECT' if pad else 'SAME'
    yr = O.conv2d(xt, wt, s, mode, bt)
    dy = rng.normal(size=tuple(yr.shape)).astype(np.float32)
    dyt = t64(dy)
    gr = list(torch.autograd.grad((yr * dyt).sum(), [xt, wt] + ([bt] if bias else [])))
    if bf16:
        name += ' bf16'
        if bf16_conv_rule(Cc, N):
            yr = O.conv2d(bf16r(xt), bf16r(wt), s, mode, bt)
        if bf16_conv_rule(N, Cc):
            gr[0] = torch.autograd.grad((O.conv2d(xt, bf16r(wt), s, mode) * bf16r(dyt)).sum(), xt)[0]
        if bf16_wgrad_rule(Cc, N):
            gr[1] = torch.autograd.grad((O.conv2d(bf16r(xt), wt, s, mode) * bf16r(dyt)).sum(), wt)[0]
    y, dx, dw, db = net.test_op('conv2d', [x, w, b, dy], [NB, H, W, Cc, N, k, s, pad],
                                [tuple(yr.shape), x.shape, w.shape, (N,) if bias else None])
    rows = [(f'conv2d[{name}] y', relerr(y, yr.detach().numpy()), FWD_TOL),
            (f'conv2d[{name}] dx', relerr(dx, gr[0].numpy()), BWD_TOL),
            (f'conv2d[{name}] dw', relerr(dw, gr[1].numpy()), BWD_TOL)]
    if bias:
        rows.append((f'conv2d[{name}] db', relerr(db, gr[2].numpy()), BWD_TOL))
    return rows


DECONV_CASES = [('g_dconv_1-like 512->256', 2, 4, 4, 512, 256), ('g_dconv_3-like 128->64', 2, 8, 8, 128, 64),
                ('g_dconv_4-like 64->3', 2, 16, 16, 64, 3), ('odd C 24->20', 1, 4, 8, 24, 20),
                ('g_dconv_2-like 256->128 8x8 (halo phases)', 4, 8, 8, 256, 128),
                ('g_dconv_3-like 128->64 32x32 (halo, 64 channels)', 8, 32, 32, 128, 64)]


def check_deconv(net, case, seed=1, bf16=False):
    name, NB, H, W, Cin, Cout = case
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(NB, H, W, Cin)).astype(np.float32)
    w = (rng.normal(size=(5, 5, Cout, Cin)) * 0.1).astype(np.float32)
    b = rng.normal(size=(Cout,)).astype(np.float32)
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    yr = O.deconv2d(xt, wt, bt, (2 * H, 2 * W))
    dy = rng.normal(size=tuple(yr.shape)).astype(np.float32)
    dyt = t64(dy)
    gr = list(torch.autograd.grad((yr * dyt).sum(), [xt, wt, bt]))
    if bf16:
        name += ' bf16'
        zb = torch.zeros_like(bt)
        if bf16_conv_rule(Cin, Cout):
            yr = O.deconv2d(bf16r(xt), bf16r(wt), bt, (2 * H, 2 * W))
        if bf16_conv_rule(Cout, Cin):
            gr[0] = torch.autograd.grad((O.deconv2d(xt, bf16r(wt), zb, (2 * H, 2 * W)) * bf16r(dyt)).sum(), xt)[0]
        if bf16_wgrad_rule(Cout, Cin):
            gr[1] = torch.autograd.grad((O.deconv2d(bf16r(xt), wt, zb, (2 * H, 2 * W)) * bf16r(dyt)).sum(), wt)[0]
    y, dx, dw, db = net.test_op('deconv2d', [x, w, b, dy], [NB, H, W, Cin, Cout],
                                [tuple(yr.shape), x.shape, w.shape, (Cout,)])
    return [(f'deconv2d[{name}] y', relerr(y, yr.detach().numpy()), FWD_TOL),
            (f'deconv2d[{name}] dx', relerr(dx, gr[0].numpy()), BWD_TOL),
            (f'deconv2d[{name}] dw', relerr(dw, gr[1].numpy()), BWD_TOL),
            (f'deconv2d[{name}] db', relerr(db, gr[2].numpy()), BWD_TOL)]


ACTS = {0: None, 1: 'relu', 2: 'lrelu', 3: 'tanh'}


def check_in_act(net, act, shape=(3, 8, 16, 128), seed=2):
    rng = np.random.RandomState(seed)
    x = (rng.normal(size=shape) * 1.7 + 0.3).astype(np.float32)
    dy = rng.normal(size=shape).astype(np.float32)
    xt = t64(x, True)
    yr = O.activation(O.instance_norm(xt), ACTS[act])
    gx, = torch.autograd.grad((yr * t64(dy)).sum(), xt)
    y, dx = net.test_op('in_act', [x, dy], list(shape) + [act], [shape, shape])
    return [(f'in_act[{ACTS[act]} {shape}] y', relerr(y, yr.detach().numpy()), FWD_TOL),
            (f'in_act[{ACTS[act]} {shape}] dx', relerr(dx, gx.numpy()), BWD_TOL)]


# every register-resident form of the tangent kernels (k_norm.hip: in_tangent_bwd_vr / in_bwd_vr<.., 1> -- H*W <= 64 / 128 / 256 /
# 512 / 1024, ragged pixel counts) and the multi-pass forms behind them (H*W > 1024); 32- and 64-channel blocks
TANGENT_SHAPES = [(2, 6, 7, 32), (1, 8, 16, 64), (2, 10, 10, 32), (1, 16, 16, 32), (1, 12, 25, 64), (1, 16, 32, 32), (1, 25, 28, 32),
                  (1, 32, 32, 32), (1, 33, 32, 32)]


def check_in_tangent(net, act=2, shape=(2, 8, 8, 64), seed=3):
    """Forward-mode tangent of IN+act and its reverse (used by the WGAN-GP second-order path)."""
    rng = np.random.RandomState(seed)
    x = (rng.normal(size=shape) * 1.3).astype(np.float32)
    tx = rng.normal(size=shape).astype(np.float32)
    dty = rng.normal(size=shape).astype(np.float32)
    xt, txt = t64(x, True), t64(tx, True)
    f = lambda u: O.activation(O.instance_norm(u), ACTS[act])
    yr, tyr = torch.autograd.functional.jvp(f, (xt,), (txt,), create_graph=True)
    gx, gtx = torch.autograd.grad((tyr * t64(dty)).sum(), [xt, txt])
    y, ty, dtx, dx = net.test_op('in_tangent', [x, tx, dty], list(shape) + [act], [shape] * 4)
    return [(f'in_tangent[{ACTS[act]} {shape}] y', relerr(y, yr.detach().numpy()), FWD_TOL),
            (f'in_tangent[{ACTS[act]} {shape}] ty', relerr(ty, tyr.detach().numpy()), FWD_TOL),
            (f'in_tangent[{ACTS[act]} {shape}] dtx', relerr(dtx, gtx.numpy()), BWD_TOL),
            (f'in_tangent[{ACTS[act]} {shape}] dx_primal', relerr(dx, gx.numpy()), BWD_TOL)]


def _bn_block(x, g, b, act):
    return O.activation(O.batch_norm_train(x, g, b)[0], ACTS[act])


def check_bn_act(net, act=2, shape=(3, 8, 8, 64), seed=6):
    """The block norm of E_norm / G_norm / D_norm = 'batch' (normalization.py:19-25) + activation: forward, dx, dgamma, dbeta.
    The shapes cross the kernels' 256-row chunks (R = 192 .. 4,608 rows) and leave partial 64-channel groups."""
    rng = np.random.RandomState(seed)
    x = (rng.normal(size=shape) * 1.3 + 0.4).astype(np.float32)
    g = (1.0 + 0.3 * rng.normal(size=shape[-1])).astype(np.float32)
    b = (0.2 * rng.normal(size=shape[-1])).astype(np.float32)
    dy = rng.normal(size=shape).astype(np.float32)
    xt, gt, bt = t64(x, True), t64(g, True), t64(b, True)
    yr = _bn_block(xt, gt, bt, act)
    gr = torch.autograd.grad((yr * t64(dy)).sum(), [xt, gt, bt])
    y, dx, dg, db = net.test_op('bn_act', [x, g, b, dy], list(shape) + [act], [shape, shape, g.shape, b.shape])
    tag = f'bn_act[{ACTS[act]} {shape}]'
    return [(f'{tag} y', relerr(y, yr.detach().numpy()), FWD_TOL), (f'{tag} dx', relerr(dx, gr[0].numpy()), BWD_TOL),
            (f'{tag} dgamma', relerr(dg, gr[1].numpy()), BWD_TOL), (f'{tag} dbeta', relerr(db, gr[2].numpy()), BWD_TOL)]


def check_bn_tangent(net, act=2, shape=(3, 8, 8, 64), seed=7):
    """Forward-mode tangent of BatchNorm + activation and its reverse (the gradient penalty through a D_norm = 'batch'
    discriminator): ty, and the gradients of <dty, ty> w.r.t. the tangent, the primal input and gamma against autograd's
    double backward of the same formula."""
    rng = np.random.RandomState(seed)
    x = (rng.normal(size=shape) * 1.3 + 0.4).astype(np.float32)
    g = (1.0 + 0.3 * rng.normal(size=shape[-1])).astype(np.float32)
    b = (0.2 * rng.normal(size=shape[-1])).astype(np.float32)
    tx = rng.normal(size=shape).astype(np.float32)
    dty = rng.normal(size=shape).astype(np.float32)
    xt, gt, bt, txt = t64(x, True), t64(g, True), t64(b, True), t64(tx, True)
    yr, tyr = torch.autograd.functional.jvp(lambda u: _bn_block(u, gt, bt, act), (xt,), (txt,), create_graph=True)
    gx, gtx, gg = torch.autograd.grad((tyr * t64(dty)).sum(), [xt, txt, gt])
    y, ty, dtx, dx, dg = net.test_op('bn_tangent', [x, g, b, tx, dty], list(shape) + [act], [shape] * 4 + [g.shape])
    tag = f'bn_tangent[{ACTS[act]} {shape}]'
    return [(f'{tag} y', relerr(y, yr.detach().numpy()), FWD_TOL), (f'{tag} ty', relerr(ty, tyr.detach().numpy()), FWD_TOL),
            (f'{tag} dtx', relerr(dtx, gtx.numpy()), BWD_TOL), (f'{tag} dx_primal', relerr(dx, gx.numpy()), BWD_TOL),
            (f'{tag} dgamma', relerr(dg, gg.numpy()), BWD_TOL)]


def check_simple(net, op, shape, extra, ref_fn, seed=4, positive=False):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=shape).astype(np.float32)
    xt = t64(x, True)
    yr = ref_fn(xt)
    dy = rng.normal(size=tuple(yr.shape)).astype(np.float32)
    gx, = torch.autograd.grad((yr * t64(dy)).sum(), xt)
    y, dx = net.test_op(op, [x, dy], list(shape) + [extra], [tuple(yr.shape), shape])
    return [(f'{op}{shape} y', relerr(y, yr.detach().numpy()), FWD_TOL),
            (f'{op}{shape} dx', relerr(dx, gx.numpy()), BWD_TOL)]


def check_bn(net, shape=(4, 4, 4, 128), seed=5):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=shape).astype(np.float32)
    g = rng.uniform(0.5, 1.5, size=(shape[3],)).astype(np.float32)
    b = rng.normal(size=(shape[3],)).astype(np.float32) * 0.3
    b = b.astype(np.float32)
    dy = rng.normal(size=shape).astype(np.float32)
    xt, gt, bt = t64(x, True), t64(g, True), t64(b, True)
    yr = torch.relu(O.batch_norm_train(xt, gt, bt)[0])
    gr = torch.autograd.grad((yr * t64(dy)).sum(), [xt, gt, bt])
    y, dx, dg, db = net.test_op('bn_relu', [x, g, b, dy], list(shape), [shape, shape, g.shape, b.shape])
    return [('bn_relu y', relerr(y, yr.detach().numpy()), FWD_TOL), ('bn_relu dx', relerr(dx, gr[0].numpy()), BWD_TOL),
            ('bn_relu dgamma', relerr(dg, gr[1].numpy()), BWD_TOL), ('bn_relu dbeta', relerr(db, gr[2].numpy()), BWD_TOL)]


def check_prelu(net, n=5000, seed=6):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(n,)).astype(np.float32)
    lk = np.array([0.2], dtype=np.float32)
    dy = rng.normal(size=(n,)).astype(np.float32)
    xt, lt = t64(x, True), t64(lk[0], True)
    yr = O.prelu(xt, lt)
    gx, gl = torch.autograd.grad((yr * t64(dy)).sum(), [xt, lt])
    y, dx, dl = net.test_op('prelu', [x, lk, dy], [n], [(n,), (n,), (1,)])
    return [('prelu y', relerr(y, yr.detach().numpy()), FWD_TOL), ('prelu dx', relerr(dx, gx.numpy()), BWD_TOL),
            ('prelu dleak', relerr(dl, gl.numpy().reshape(1)), BWD_TOL)]


def check_gate(net, n=4096, seed=7):
    rng = np.random.RandomState(seed)
    h, rg, im, dy = [rng.normal(size=(n,)).astype(np.float32) for _ in range(4)]
    y, dh, drg, dim = net.test_op('gate', [h, rg, im, dy], [n], [(n,)] * 4)
    return [('gate y', relerr(y, h + rg * im), FWD_TOL), ('gate dh', relerr(dh, dy), BWD_TOL),
            ('gate drg', relerr(drg, dy * im), BWD_TOL), ('gate dimg', relerr(dim, dy * rg), BWD_TOL)]


def check_linear1(net, NB=6, K=4096, seed=8):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(NB, K)).astype(np.float32)
    w = (rng.normal(size=(K, 1)) * 0.05).astype(np.float32)
    b = np.array([0.3], dtype=np.float32)
    dy = rng.normal(size=(NB, 1)).astype(np.float32)
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    yr = O.linear(xt, wt, bt)
    gr = torch.autograd.grad((yr * t64(dy)).sum(), [xt, wt, bt])
    y, dx, dw, db = net.test_op('linear1', [x, w, b, dy], [NB, K], [(NB, 1), x.shape, w.shape, (1,)])
    return [('linear1 y', relerr(y, yr.detach().numpy()), FWD_TOL), ('linear1 dx', relerr(dx, gr[0].numpy()), BWD_TOL),
            ('linear1 dw', relerr(dw, gr[1].numpy()), BWD_TOL), ('linear1 db', relerr(db, gr[2].numpy()), BWD_TOL)]


def check_sn(net, K, Co, seed=9):
    rng = np.random.RandomState(seed)
    W = (rng.normal(size=(K, Co)) * 0.02).astype(np.float32)
    u = np.clip(rng.normal(size=(1, Co)), -2, 2).astype(np.float32)
    dWb = rng.normal(size=(K, Co)).astype(np.float32)
    Wt = t64(W, True)
    Wbr = O.spectral_normed_weight(Wt, t64(u))
    gW, = torch.autograd.grad((Wbr * t64(dWb)).sum(), Wt)
    Wbar, dW, sigma = net.test_op('sn', [W, u, dWb], [K, Co], [(K, Co), (K, Co), (1,)])
    return [(f'sn[{K}x{Co}] Wbar', relerr(Wbar, Wbr.detach().numpy()), FWD_TOL),
            (f'sn[{K}x{Co}] dW', relerr(dW, gW.numpy()), BWD_TOL)]


def check_acgan(net, B=16, NC=14, seed=10):
    rng = np.random.RandomState(seed)
    lg = (rng.normal(size=(B, NC)) * 2).astype(np.float32)
    lab = rng.randint(0, NC, size=(B,))
    rows = []
    for mode in (0, 1):
        lt = t64(lg, True)
        lg_, ld_ = O.acgan_loss_focal(lt, torch.tensor(lab), lt)
        ref = ld_ if mode == 0 else lg_
        gl, = torch.autograd.grad(ref, lt)
        loss, dl = net.test_op('acgan', [lg, lab.astype(np.float32)], [B, NC, mode], [(1,), (B, NC)])
        rows += [(f'acgan mode{mode} loss', relerr(loss, ref.detach().numpy().reshape(1)), FWD_TOL),
                 (f'acgan mode{mode} dlogits', relerr(dl, gl.numpy()), BWD_TOL)]
    return rows


def check_gp(net, B=4, per=3000, seed=11):
    rng = np.random.RandomState(seed)
    g = rng.normal(size=(B, per)).astype(np.float32) * 0.05
    gt = t64(g, True)
    nr = torch.sqrt((gt ** 2).sum(dim=1))
    gp = net.config.lambda_gp * torch.mean((nr - 1) ** 2)
    vr, = torch.autograd.grad(gp, gt)
    v, norms, gpv = net.test_op('gp', [g], [B, per], [(B, per), (B,), (1,)])
    return [('gp v', relerr(v, vr.numpy()), BWD_TOL), ('gp norms', relerr(norms, nr.detach().numpy()), FWD_TOL),
            ('gp value', relerr(gpv, gp.detach().numpy().reshape(1)), FWD_TOL)]


def check_rmsprop(net, n=10000, seed=12):
    rng = np.random.RandomState(seed)
    w, g = rng.normal(size=(n,)).astype(np.float32), rng.normal(size=(n,)).astype(np.float32)
    ms = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
    mom = np.zeros(n, dtype=np.float32)
    wt, gt, mst, momt = t64(w), t64(g), t64(ms), t64(mom)
    O.rmsprop_apply(wt, gt, mst, momt, 2e-4)
    w2, ms2, mom2 = net.test_op('rmsprop', [w, g, ms, mom, np.array([2e-4], np.float32)], [n], [(n,)] * 3)
    return [('rmsprop w', relerr(w2, wt.numpy()), 1e-6), ('rmsprop ms', relerr(ms2, mst.numpy()), 1e-6),
            ('rmsprop mom', relerr(mom2, momt.numpy()), 1e-5)]


def all_op_checks(net):
    rows = []
    for c in CONV_CASES:
        rows += check_conv(net, c)
    for c in DECONV_CASES:
        rows += check_deconv(net, c)
    for a in (0, 1, 2):
        rows += check_in_act(net, a)
    rows += check_in_act(net, 2, shape=(2, 4, 8, 512))
    # maps of more than 1,024 pixels: the chunked two-stage InstanceNorm kernels (k_norm.hip; 256-row chunks, a partial last chunk,
    # a partial 64-channel group) -- forward, backward, forward tangent and its reverse
    rows += check_in_act(net, 1, shape=(2, 48, 40, 96), seed=21)
    rows += check_in_act(net, 2, shape=(3, 33, 32, 64), seed=22)
    rows += check_in_tangent(net, 2, shape=(2, 40, 36, 64), seed=23)
    rows += check_in_tangent(net, 2, shape=(1, 64, 33, 160), seed=24)
    rows += check_in_tangent(net, 2)
    rows += check_in_tangent(net, 2, shape=(2, 4, 8, 128), seed=33)
    for i, shape in enumerate(TANGENT_SHAPES):
        rows += check_in_tangent(net, 2, shape=shape, seed=40 + i)
    rows += check_simple(net, 'act', (2, 4, 4, 32), 3, lambda t: torch.tanh(t))
    rows += check_simple(net, 'act', (2, 4, 4, 32), 2, lambda t: O.activation(t, 'lrelu'))
    rows += check_simple(net, 'resize', (2, 64, 64, 3), 128, lambda t: O.resize_bicubic(t, 128))
    rows += check_simple(net, 'pool2', (2, 8, 8, 24), 0, lambda t: O.avg_pool2(t))
    rows += check_simple(net, 'gmean', (3, 2, 2, 64), 1, lambda t: O.avg_pool_same(torch.relu(t), 8).reshape(3, 1, 1, 64))
    rows += check_simple(net, 'gmean', (3, 4, 4, 96), 0, lambda t: t.mean(dim=(1, 2), keepdim=True))

    def mm(t):
        mn, mx = t.amin(dim=(1, 2), keepdim=True), t.amax(dim=(1, 2), keepdim=True)
        return (t - mn) / (mx - mn)
    rows += check_simple(net, 'minmax', (2, 8, 8, 40), 0, mm)
    # maps of more than 4,096 pixels: the chunked min-max gate (k_elem.hip; 8-channel and 64-channel block forms, partial chunks)
    rows += check_simple(net, 'minmax', (2, 80, 64, 40), 0, mm, seed=41)
    rows += check_simple(net, 'minmax', (1, 72, 65, 64), 0, mm, seed=42)
    rows += check_bn(net)
    for act, shape in ((1, (3, 8, 8, 64)), (2, (2, 16, 32, 128)), (0, (3, 24, 16, 96)), (2, (64, 4, 8, 512)), (1, (2, 3, 32, 40))):
        rows += check_bn_act(net, act, shape)
    for act, shape in ((2, (3, 8, 8, 64)), (2, (2, 16, 32, 128)), (2, (5, 16, 16, 256)), (1, (2, 3, 32, 40))):
        rows += check_bn_tangent(net, act, shape)
    rows += check_prelu(net)
    rows += check_gate(net)
    rows += check_linear1(net)
    rows += check_sn(net, 99, 8)
    rows += check_sn(net, 1152, 128)
    rows += check_acgan(net)
    rows += check_gp(net)
    rows += check_rmsprop(net)
    return rows


# ------------------------------------------------------------------------------------------------
# network forward checks (full-size architectures, small batch)
# ------------------------------------------------------------------------------------------------
def net_checks(net: ea.EdgeGAN, values, seed=20, bf16=False, tol=5e-4):
    """bf16=True: engine in compute_dtype='bf16' against the oracle under O.bf16_operands() (same operand rounding,
    exact accumulation), so the forward tolerance stays the fp32 one."""
    with O.bf16_operands(bf16):
        return _net_checks(net, values, seed, tol)


def _net_checks(net, values, seed, tol):
    cfg = net.config
    ocfg = oracle_config(cfg)
    p = {k: t64(v) for k, v in values.items()}
    rng = np.random.RandomState(seed)
    NB = cfg.batch_size
    rows = []
    g_in = cfg.z_dim + (cfg.num_classes if cfg.multiclasses else 0)
    zin = rng.normal(size=(NB, g_in)).astype(np.float32)
    for g in ('G1', 'G2'):
        ref = O.generator(p, g, t64(zin), ocfg).numpy()
        out = net.net_forward(g, zin, ref.shape)
        rows.append((f'net {g} forward', relerr(out, ref), tol))
    x = rng.uniform(-1, 1, size=(NB, cfg.height, cfg.width, 3)).astype(np.float32)
    ref = O.discriminator(p, 'D', t64(x))[1].numpy()
    rows.append(('net D forward', relerr(net.net_forward('D', x, ref.shape), ref), tol))
    xp = rng.uniform(-1, 1, size=(NB, cfg.dis_size, cfg.dis_size, 3)).astype(np.float32)
    for d in ('D_patch2', 'D_patch3'):
        ref = O.discriminator(p, d, t64(xp))[1].numpy()
        rows.append((f'net {d} forward', relerr(net.net_forward(d, xp, ref.shape), ref), tol))
    xe = rng.uniform(-1, 1, size=(NB, cfg.height, cfg.width // 2, 3)).astype(np.float32)
    ref = O.encoder(p, t64(xe), 0.37, ocfg)[0].numpy()
    rows.append(('net E forward', relerr(net.net_forward('E', xe, (NB, 1, 1, cfg.z_dim), eps=0.37).reshape(ref.shape), ref), tol))
    if cfg.multiclasses:
        ref = O.classifier(p, t64(xe), ocfg).numpy()
        rows.append(('net D2 forward', relerr(net.net_forward('D2', xe, (NB, 1, 1, cfg.num_classes)).reshape(ref.shape), ref), tol))
    return rows


# ------------------------------------------------------------------------------------------------
# sub-step gradient + update checks
# ------------------------------------------------------------------------------------------------
def substep_checks(net: ea.EdgeGAN, values, batch, grad_tol=1e-2, iterations=1, bf16=False, loss_tol=1e-4,
                   self_consistency=None, share_oracle=False):
    """bf16=True (engine created with compute_dtype='bf16'): the oracle runs under O.bf16_operands(), i.e. with the
    engine's operand rounding in all three maps of every qualifying contraction.  What remains between the two is
    (a) fp32-vs-fp64 noise in the values that GET rounded -- ~1e-4 of them land on the other side of a bf16 rounding
    boundary, a 2^-8 relative change of that element -- and (b) the different evaluation order of the gradient-
    penalty second derivative.  (a) is a 1e-4..1e-3 forward perturbation, which this ill-conditioned graph
    amplifies ~100x in some gradients (e_optim through the 4x4 / 2x2 InstanceNorms): an fp32 evaluation of the
    ORACLE'S OWN emulation deviates from its fp64 evaluation by a median 0.3 in e_optim (tests/test_bf16_gpu.py has
    the numbers).  The gradient tolerance is therefore set per variable from that measured self-consistency:
    max(grad_tol, 3 x ||g_oracle_fp32 - g_oracle_fp64|| / ||g_oracle_fp64||).
    self_consistency=True applies the same rule to an fp32 engine (used where a fixture is known to sit on a ReLU /
    L1 kink: the oracle's own fp32 run then deviates from its fp64 run by the same few percent the engine does)."""
    with O.bf16_operands(bf16):
        return _substep_checks(net, values, batch, grad_tol, iterations, loss_tol,
                               self_consistency=bf16 if self_consistency is None else self_consistency, share_oracle=share_oracle)


# The oracle's side of _substep_checks does not depend on the engine (before every sub-step the engine is put on the
# oracle's state), so tests that run the same fixture in two compute modes (tests/test_step_gpu.py and their f32x3 re-runs in
# tests/test_x3_gpu.py) share one oracle trace: computed by the first, consumed -- and dropped -- by the second.
_ORACLE_TRACES = {}


def _trace_key(cfg, values, batch, iterations, self_consistency):
    import hashlib
    h = hashlib.sha1(repr((oracle_config(cfg), iterations, bool(self_consistency), bool(O._BF16_OPERANDS))).encode())
    for d in (values, batch):
        for k in sorted(d):
            h.update(k.encode())
            h.update(np.ascontiguousarray(d[k]).tobytes())
    return h.hexdigest()


def _oracle_trace(cfg, values, batch, iterations, self_consistency):
    """Per sub-step: the fp64 gradients and losses at the (fp32-representable) state the step starts from, the optional
    per-variable tolerance from the oracle's own fp32 run, the weights / RMS slots before the update and the state every
    optimised variable (and its two slots) is put on afterwards; at the end the generators' BatchNorm moving statistics."""
    ocfg = oracle_config(cfg)
    om = O.OracleModel(ocfg, values, dtype=DT)
    inp = O.StepInputs(t64(batch['images']), t64(batch['z']), t64(batch['alpha']), t64(batch['eps']))
    if self_consistency:
        om32 = O.OracleModel(ocfg, values, dtype=torch.float32)
        inp32 = O.StepInputs(*[torch.tensor(np.asarray(batch[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
    steps = []
    for it in range(iterations):
        for k, name in enumerate(O.SUBSTEPS, start=1):
            if name not in om.substep_names():      # register_optim_if: d_optim2 / d_optim_patch2 / d_optim_patch3 behind their flags
                continue
            grads, losses = om.compute_grads(name, inp, inp.eps[0])
            var_tol = {}
            if self_consistency:
                for d32, d64 in ((om32.p, om.p), (om32.ms, om.ms), (om32.mom, om.mom)):
                    for vn in d32:
                        d32[vn].copy_(d64[vn].float())
                g32, _ = om32.compute_grads(name, inp32, inp32.eps[0])
                for vn, g in grads.items():
                    var_tol[vn] = 3.0 * rel_l2(g32[vn].double().numpy(), g.numpy(), 1e-5 * float(g.abs().max()) + 1e-300)
            before = {vn: om.p[vn].numpy().astype(np.float32) for vn in grads}          # (fp32-representable by construction)
            ms_before = {vn: om.ms[vn].numpy().astype(np.float32) for vn in grads}
            om.apply_grads(grads)
            om.update_bn_moving(om._GENS[name])
            post = {}
            for vn in grads:
                for d in (om.p, om.ms, om.mom):
                    d[vn].copy_(d[vn].float().double())
                post[vn] = (om.p[vn].numpy().astype(np.float32), om.ms[vn].numpy().astype(np.float32),
                            om.mom[vn].numpy().astype(np.float32))
            steps.append(dict(it=it, k=k, name=name, grads={vn: g.numpy().copy() for vn, g in grads.items()},
                              losses={lk: float(v) for lk, v in losses.items()}, var_tol=var_tol, before=before,
                              ms_before=ms_before, post=post))
    bn = {}
    for g in ('G1', 'G2'):
        for sname in ('moving_mean', 'moving_variance'):
            vn = f'{g}/batch_norm/BatchNorm/{sname}'
            bn[vn] = om.p[vn].numpy().copy()
    return dict(steps=steps, bn=bn)


def _substep_checks(net, values, batch, grad_tol, iterations, loss_tol, self_consistency=False, share_oracle=False):
    """The 7 sub-steps in update_model order.  Before each sub-step the engine is put on exactly the
    oracle's state (oracle values are kept fp32-representable), because the GAN trajectory at random
    init is chaotic: an fp32-vs-fp64 run of the ORACLE ITSELF drifts to 1e-1 relative gradient
    differences by sub-step 7 (measured; DESIGN.md "conditioning"), which says nothing about kernels.
    Per sub-step: every optimised variable's gradient vs oracle autograd (fp64, incl. the WGAN-GP
    double backward), the reported loss, and that RMSProp was applied to the variable and its slot.
    Gradient metric: relative L2 error, tolerance 1e-2.  Why not the tight max-norm of the op tests:
    ReLU / LeakyReLU / PReLU derivatives are discontinuous at 0, and among the ~1e6 activations of the
    small-extent layers (4x4 .. 16x16) one or two pre-activations per sub-step land within fp32
    rounding (|y| < 1e-6) of the kink; whichever side an fp32 implementation rounds to changes that
    channel's gradient by ~1/(B*H*W) (percents at B=2..4) -- the fp32 run of the ORACLE ITSELF flips
    such masks against its fp64 run (measured, DESIGN.md "conditioning").  A flip moves the L2 error
    of the affected variables to ~1e-3; wiring / formula bugs move it to >= 1e-1.  When no flip
    occurs the engine agrees with fp64 autograd to ~2e-5 max-norm (the value in each row's name);
    smooth-regime accuracy is pinned separately by the op tests at 5e-5 max-norm."""
    cfg = net.config
    # share_oracle: this fixture is run twice in one session (exact-fp32 and f32x3): the first run leaves its oracle trace for
    # the second, which consumes it
    key = _trace_key(cfg, values, batch, iterations, self_consistency) if share_oracle else None
    trace = _ORACLE_TRACES.pop(key, None) if share_oracle else None
    if trace is None:
        trace = _oracle_trace(cfg, values, batch, iterations, self_consistency)
        if share_oracle:
            _ORACLE_TRACES[key] = trace
    net.set_variables(values)
    net.upload_batch(batch['images'], batch['z'], batch['alpha'], batch['eps'])
    rows = []
    loss_map = {1: ['joint_dis_dloss'], 2: ['image_dis_dloss'], 3: ['edge_dis_dloss'], 4: ['loss_d_ac'],
                5: ['edge_gloss', 'image_gloss'], 6: ['zl_loss'], 7: ['edge_gloss', 'image_gloss']}
    for st in trace['steps']:
        k, name, grads, losses, var_tol = st['k'], st['name'], st['grads'], st['losses'], st['var_tol']
        tag = f"it{st['it']} substep {k} {name}"
        net.substep(k, apply=False)
        gmax = max(float(np.abs(g).max()) for g in grads.values())
        hip_g = {}
        for vn, g in grads.items():
            hip_g[vn] = net.get_variable(vn + '/grad')
            if float(np.abs(g).max()) < 1e-9 * gmax:
                # structurally zero gradient (a bias in front of an InstanceNorm / min-max gate):
                # fp32 may only leave rounding noise, bounded by 1e-5 of the sub-step's largest gradient
                rows.append((f'{tag} grad {vn} (structural zero: noise/1e-5*gmax)',
                             float(np.abs(hip_g[vn]).max()) / (1e-5 * gmax), 1.0))
            else:
                mx = relerr(hip_g[vn], g, 1e-5 * gmax)
                rows.append((f'{tag} grad {vn} [L2; max-norm {mx:.1e}; |g|/gmax {float(np.abs(g).max()) / gmax:.0e}]',
                             rel_l2(hip_g[vn], g, 1e-5 * gmax), max(grad_tol, var_tol.get(vn, 0.0))))
        hl = net.losses()
        for lk in loss_map[k]:
            ref = losses[lk]
            rows.append((f'{tag} loss {lk}', abs(hl[lk] - ref) / (abs(ref) + 1e-30), loss_tol))
        # the optimizer: the engine must apply tf RMSProp (Appendix B.7) with ITS gradient to exactly
        # these variables and slots (gradient parity is the rows above; RMSProp's g/sqrt(ms) would
        # otherwise re-weight element errors by the 1e3 dynamic range of GAN gradients at init)
        net.substep(k, apply=True)
        for vn in grads:
            before, ms_before = st['before'][vn].astype(np.float64), st['ms_before'][vn].astype(np.float64)
            gh = hip_g[vn].astype(np.float64).reshape(before.shape)
            ms_new = ms_before + (gh * gh - ms_before) * (1.0 - 0.9)
            w_new = before - cfg.learning_rate * gh / np.sqrt(ms_new + 1e-10)
            got = net.get_variable(vn).astype(np.float64)
            ulp = float(np.spacing(np.float32(np.abs(w_new).max())))
            e = 0.0 if np.max(np.abs(got - w_new)) <= 2 * ulp else relerr(got, w_new)
            rows.append((f'{tag} rmsprop-applied {vn}', e, 1e-5))
            rows.append((f'{tag} rms-slot {vn}', relerr(net.get_variable(vn + '/RMSProp'), ms_new), 1e-5))
        for vn, (pw, pms, pmom) in st['post'].items():      # put the engine on the oracle's (fp32-representable) state
            net.set_variable(vn, pw)
            net.set_variable(vn + '/RMSProp', pms)
            net.set_variable(vn + '/RMSProp_1', pmom)
    for vn, ref in trace['bn'].items():
        rows.append((f'bn {vn}', relerr(net.get_variable(vn), ref), 1e-4))
    return rows


def format_rows(rows, only_fail=False):
    out = []
    for n, e, t in rows:
        ok = e <= t
        if only_fail and ok:
            continue
        out.append(f"{'ok  ' if ok else 'FAIL'} {e:10.3e} (tol {t:.0e})  {n}")
    return '\n'.join(out)


def batch_sweep_checks(net, batches=(1, 3, 4, 5, 8)):
    """Every batch-dependent op at several batch sizes (shapes of the encoder's top blocks, the
    discriminator tail and the generator head)."""
    rows = []
    for nb in batches:
        for case in [('E top 3x3 REFLECT 4x4 C256->256', nb, 4, 4, 256, 256, 3, 1, 1, True),
                     ('E top 1x1 REFLECT 4x4 C128->256', nb, 4, 4, 128, 256, 1, 1, 1, True),
                     ('E 3x3 REFLECT 8x8 C128->128', nb, 8, 8, 128, 128, 3, 1, 1, True),
                     ('D tail 4x4s2 8x8 C128->256', nb, 8, 8, 128, 256, 4, 2, 0, False),
                     ('stem 4x4s2 C3->64', nb, 16, 16, 3, 64, 4, 2, 0, True),
                     ('FC8 linear 512->100', nb, 1, 1, 512, 100, 1, 1, 0, True),
                     ('g_lin linear 114->1024', nb, 1, 1, 114, 1024, 1, 1, 0, True)]:
            rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_conv(net, case, seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_deconv(net, ('dconv 4x4 C128->64', nb, 4, 4, 128, 64), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_in_act(net, 1, shape=(nb, 4, 4, 256), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_in_tangent(net, 2, shape=(nb, 4, 8, 128), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_bn(net, shape=(nb, 4, 4, 128), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_linear1(net, NB=nb, K=2048, seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in
                 check_simple(net, 'pool2', (nb, 4, 4, 256), 0, lambda t: O.avg_pool2(t), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in
                 check_simple(net, 'gmean', (nb, 2, 2, 256), 1,
                              lambda t: O.avg_pool_same(torch.relu(t), 8).reshape(t.shape[0], 1, 1, 256), seed=nb)]
        rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in
                 check_simple(net, 'resize', (nb, 16, 16, 3), 32, lambda t: O.resize_bicubic(t, 32), seed=nb)]
    return rows


def real_shape_checks(net, nbs=(4, 8), bf16=False):
    rows = []
    for nb in nbs:
        for case in [('E 256_2/res1 16x16 C128->256', nb, 16, 16, 128, 256, 3, 1, 1, True),
                     ('E 512_4/res2 4x4 C512->512', nb, 4, 4, 512, 512, 3, 1, 1, True),
                     ('E 128_1/res1 32x32 C64->128', nb, 32, 32, 64, 128, 3, 1, 1, True),
                     ('Dp conv_4 16x16 C256->512', nb, 16, 16, 256, 512, 4, 2, 0, False),
                     ('Dp conv_3 32x32 C128->256', nb, 32, 32, 128, 256, 4, 2, 0, False)]:
            rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_conv(net, case, seed=nb, bf16=bf16)]
        for case in [('g_dconv_1 4x4 512->256', nb, 4, 4, 512, 256), ('g_dconv_2 8x8 256->128', nb, 8, 8, 256, 128),
                     ('g_dconv_3 16x16 128->64', nb, 16, 16, 128, 64)]:
            rows += [(f'NB={nb} ' + r[0], r[1], r[2]) for r in check_deconv(net, case, seed=nb, bf16=bf16)]
    return rows


LONG_M_CASES = [
    # many pixels per wgrad block: exercises the multi-block pixel-offset ring of the bf16 wgrad (> 16 K steps per
    # block) and the deep split reductions, at shapes the batch-2 network tests never reach
    ('long-M 3x3 C64->64', 16, 64, 64, 64, 64, 3, 1, 0, True),
    ('long-M 4x4s2 C32->128', 8, 128, 128, 32, 128, 4, 2, 0, False),
    ('long-M REFLECT 3x3 C32->32', 4, 96, 96, 32, 32, 3, 1, 1, True),
]


def long_m_checks(net, bf16=False):
    rows = []
    for case in LONG_M_CASES:
        rows += check_conv(net, case, seed=31, bf16=bf16)
    return rows


def bf16_op_checks(net):
    """Every conv / deconv case of the fp32 op tests, on an engine created with compute_dtype='bf16'."""
    rows = []
    for case in CONV_CASES:
        rows += check_conv(net, case, bf16=True)
    for case in DECONV_CASES:
        rows += check_deconv(net, case, bf16=True)
    return rows + real_shape_checks(net, bf16=True) + long_m_checks(net, bf16=True)


# ------------------------------------------------------------------------------------------------
# round 5: the fused forms of the MRU unit (conv + PReLU as a second epilogue store, K-split slabs summed by their consumer,
# pool2_add's second store) and the input-gradient slabs read by in_bwd / pool2_bwd -- each through a composite test op that
# builds the same op sequence classifier_forward / disc_forward build (capi.cpp: mru_tail, mru_gate, act_conv)
# ------------------------------------------------------------------------------------------------
def _pool_nhwc(t):
    return O._nhwc(O.mean_pool_nchw(O._nchw(t)))


class _slope_kappa:
    """Records sum_i |t_i| of every scalar PReLU slope's gradient terms while the oracle differentiates (O._SLOPE_TERMS); a
    slope's gradient is ONE cancelling sum, good to eps x kappa relative to itself, kappa = sum|t_i| / |sum t_i| -- the rule of
    tests/test_tight_gpu.py (1e-5 x kappa)."""
    def __enter__(self):
        O._SLOPE_TERMS = {}
        return self

    def __exit__(self, *exc):
        self.terms = O._SLOPE_TERMS
        O._SLOPE_TERMS = None

    def tol(self, leak_tensor, grad, base):
        l1 = self.terms.get(id(leak_tensor), 0.0)
        return max(base, 1e-5 * l1 / max(abs(float(grad)), 1e-300))


def check_mru_tail(net, NB, H, W, C1, C2, seed=0, bf16=False, tol=(FWD_TOL, BWD_TOL)):
    rng = np.random.RandomState(seed)
    sc = 1.0 / np.sqrt(9.0)
    a = dict(x=rng.normal(size=(NB, H, W, C1)), w1=rng.normal(size=(3, 3, C1, C2)) * sc / np.sqrt(C1), b1=rng.normal(size=(C2,)) * 0.1,
             l1=np.array([0.2 + 0.1 * rng.rand()]), w2=rng.normal(size=(3, 3, C2, C2)) * sc / np.sqrt(C2), b2=rng.normal(size=(C2,)) * 0.1,
             bp=rng.normal(size=(NB, H // 2, W // 2, C2)), l2=np.array([0.15 + 0.1 * rng.rand()]))
    a = {k: v.astype(np.float32) for k, v in a.items()}
    dh = rng.normal(size=(NB, H // 2, W // 2, C2)).astype(np.float32)
    dhp = rng.normal(size=(NB, H // 2, W // 2, C2)).astype(np.float32)
    t = {k: t64(v, True) for k, v in a.items()}
    with O.bf16_operands(bf16), _slope_kappa() as sk:
        c1 = O.prelu(O.conv2d(t['x'], t['w1'], 1, 'SAME', t['b1']), t['l1'])
        c2 = O.conv2d(c1, t['w2'], 1, 'SAME', t['b2'])
        h = _pool_nhwc(c2) + t['bp']
        hp = O.prelu(h, t['l2'])
        names = ['x', 'w1', 'b1', 'l1', 'w2', 'b2', 'bp', 'l2']
        gr = torch.autograd.grad((h * t64(dh)).sum() + (hp * t64(dhp)).sum(), [t[n] for n in names])
    shapes = [h.shape, hp.shape] + [a[n].shape for n in names]
    out = net.test_op('mru_tail', [a[n] for n in names] + [dh, dhp], [NB, H, W, C1, C2], [tuple(s) for s in shapes])
    tag = f'mru_tail[{NB}x{H}x{W} {C1}->{C2}{" bf16" if bf16 else ""}]'
    rows = [(f'{tag} h', relerr(out[0], h.detach().numpy()), tol[0]), (f'{tag} prelu(h)', relerr(out[1], hp.detach().numpy()), tol[0])]
    for i, n in enumerate(names):
        rows.append((f'{tag} d{n}', relerr(out[2 + i], gr[i].detach().numpy()), sk.tol(t[n], gr[i], tol[1]) if n in ('l1', 'l2') else tol[1]))
    # the same graph with its parameters frozen (the classifier inside the generator sub-steps): only the input gradients are formed,
    # Conv_1's PReLU backward rides in Conv_2's input-gradient epilogue where that launch is un-split
    frozen = [tuple(s) if n in ('x', 'bp') else None for s, n in zip(shapes[2:], names)]
    out = net.test_op('mru_tail', [a[n] for n in names] + [dh, dhp], [NB, H, W, C1, C2, 1], [tuple(h.shape), tuple(hp.shape)] + frozen)
    for i, n in enumerate(names):
        if n in ('x', 'bp'):
            rows.append((f'{tag} frozen d{n}', relerr(out[2 + i], gr[i].detach().numpy()), tol[1]))
    return rows


def check_mru_gate(net, NB, H, W, Ch, N, seed=0, bf16=False, tol=(FWD_TOL, BWD_TOL)):
    rng = np.random.RandomState(seed)
    sc = 1.0 / 3.0
    a = dict(h=rng.normal(size=(NB, H, W, Ch)), xin=rng.normal(size=(NB, H, W, 3)), l0=np.array([0.2 + 0.1 * rng.rand()]),
             wh=rng.normal(size=(3, 3, Ch, Ch)) * sc / np.sqrt(Ch), wx=rng.normal(size=(3, 3, 3, Ch)) * sc / np.sqrt(3.0),
             bg=rng.normal(size=(Ch,)) * 0.1 + 0.5, img=rng.normal(size=(NB, H, W, Ch)), l1=np.array([0.2 + 0.1 * rng.rand()]),
             w1=rng.normal(size=(3, 3, Ch, N)) * sc / np.sqrt(Ch))
    a = {k: v.astype(np.float32) for k, v in a.items()}
    dy = rng.normal(size=(NB, H, W, N)).astype(np.float32)
    t = {k: t64(v, True) for k, v in a.items()}
    with O.bf16_operands(bf16), _slope_kappa() as sk:
        hp = O.prelu(t['h'], t['l0'])
        rg = O.conv2d(t['xin'], t['wx'], 1, 'SAME', t['bg']) + O.conv2d(hp, t['wh'], 1, 'SAME')
        rg = O.lrelu_leak(rg)
        mn, mx = rg.amin(dim=(1, 2), keepdim=True), rg.amax(dim=(1, 2), keepdim=True)
        rg = (rg - mn) / (mx - mn)
        hin = O.prelu(t['h'] + rg * t['img'], t['l1'])
        y = O.conv2d(hin, t['w1'], 1, 'SAME')
        names = ['h', 'xin', 'img', 'l0', 'wh', 'wx', 'bg', 'l1', 'w1']
        gr = torch.autograd.grad((y * t64(dy)).sum(), [t[n] for n in names])
    ins = [a['h'], a['xin'], a['l0'], a['wh'], a['wx'], a['bg'], a['img'], a['l1'], a['w1'], dy]
    out = net.test_op('mru_gate', ins, [NB, H, W, Ch, N], [tuple(y.shape)] + [a[n].shape for n in names])
    tag = f'mru_gate[{NB}x{H}x{W} {Ch}->{N}{" bf16" if bf16 else ""}]'
    rows = [(f'{tag} y', relerr(out[0], y.detach().numpy()), tol[0])]
    for i, n in enumerate(names):
        rows.append((f'{tag} d{n}', relerr(out[1 + i], gr[i].detach().numpy()), sk.tol(t[n], gr[i], tol[1]) if n in ('l0', 'l1') else tol[1]))
    return rows


def check_act_conv(net, NB, H, W, Cc, N, k, s, kind, seed=0, bf16=False, tol=(FWD_TOL, BWD_TOL)):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(NB, H, W, Cc)).astype(np.float32)
    w = (rng.normal(size=(k, k, Cc, N)) / (k * np.sqrt(Cc))).astype(np.float32)
    xt, wt = t64(x, True), t64(w, True)
    with O.bf16_operands(bf16):
        m = O.lrelu_leak(O.instance_norm(xt)) if kind == 0 else O.avg_pool2(xt)
        y = O.conv2d(m, wt, s, 'SAME')
        dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
        gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, wt])
    out = net.test_op('act_conv', [x, w, dy], [NB, H, W, Cc, N, k, s, kind], [tuple(y.shape), x.shape, w.shape])
    tag = f'{"in_act" if kind == 0 else "pool2"}->conv[{NB}x{H}x{W} {Cc}->{N} k{k}s{s}{" bf16" if bf16 else ""}]'
    return [(f'{tag} y', relerr(out[0], y.detach().numpy()), tol[0]), (f'{tag} dx', relerr(out[1], gr[0].detach().numpy()), tol[1]),
            (f'{tag} dw', relerr(out[2], gr[1].detach().numpy()), tol[1])]


def check_d_head(net, NB, H, W, N1, N2, seed=0, bf16=False, tol=(FWD_TOL, BWD_TOL)):
    """d_conv_0 + LeakyReLU + d_conv_1: the activation's backward in the second convolution's input-gradient epilogue (un-split
    halo kernel) or as the in-place pass (every other kernel) -- capi.cpp: d_head."""
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(NB, H, W, 3)).astype(np.float32)
    w0 = (rng.normal(size=(4, 4, 3, N1)) / np.sqrt(48.0)).astype(np.float32)
    w1 = (rng.normal(size=(4, 4, N1, N2)) / (4 * np.sqrt(N1))).astype(np.float32)
    xt, w0t, w1t = t64(x, True), t64(w0, True), t64(w1, True)
    with O.bf16_operands(bf16):
        y = O.conv2d(O.lrelu_leak(O.conv2d(xt, w0t, 2, 'SAME')), w1t, 2, 'SAME')
        dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
        gr = torch.autograd.grad((y * t64(dy)).sum(), [xt, w0t, w1t])
    out = net.test_op('d_head', [x, w0, w1, dy], [NB, H, W, N1, N2], [tuple(y.shape), x.shape, w0.shape, w1.shape])
    tag = f'd_head[{NB}x{H}x{W} 3->{N1}->{N2}{" bf16" if bf16 else ""}]'
    return [(f'{tag} y', relerr(out[0], y.detach().numpy()), tol[0])] + \
           [(f'{tag} d{n}', relerr(out[1 + i], gr[i].detach().numpy()), tol[1]) for i, n in enumerate(['x', 'w0', 'w1'])]


def fused_chain_checks(net, bf16=False):
    """Shapes chosen so that the K-split launches (few output tiles: small maps, many channels) and the un-split epilogue
    forms (16x16 / 32x32 maps) are both hit in the f32x3 and bf16 modes; the exact-fp32 mode runs the same graphs on its own
    kernels (every fusion falls back to the separate pass where a kernel has no such epilogue)."""
    tol = (2e-2, 5e-2) if bf16 else (FWD_TOL, 2 * BWD_TOL)   # (bf16: a rounding-boundary flip of an intermediate costs 2^-8 of that element)
    rows = []
    for (NB, H, W, C1, C2) in [(2, 8, 8, 256, 256), (4, 16, 16, 128, 128), (3, 8, 8, 512, 768), (2, 32, 32, 128, 256), (2, 16, 16, 8, 128)]:
        rows += check_mru_tail(net, NB, H, W, C1, C2, seed=C1 + NB, bf16=bf16, tol=tol)
    for (NB, H, W, Ch, N) in [(2, 8, 8, 256, 256), (4, 16, 16, 128, 128), (2, 8, 8, 512, 512), (2, 16, 16, 8, 32)]:
        rows += check_mru_gate(net, NB, H, W, Ch, N, seed=Ch + NB, bf16=bf16, tol=tol)
    for (NB, H, W, Cc, N, k, s, kind) in [(4, 16, 16, 256, 512, 4, 2, 0), (2, 8, 8, 512, 512, 3, 1, 0), (2, 32, 32, 128, 256, 4, 2, 0),
                                            (2, 8, 8, 512, 768, 1, 1, 1), (2, 16, 16, 256, 512, 1, 1, 1)]:
        rows += check_act_conv(net, NB, H, W, Cc, N, k, s, kind, seed=Cc + k, bf16=bf16, tol=tol)
    # first writer with / without the activation-backward epilogue: few pixels (K-split or 64x64 tiles) and enough for the
    # un-split halo kernel
    for (NB, H, W, N1, N2) in [(2, 32, 32, 64, 128), (16, 64, 64, 64, 128), (4, 64, 128, 64, 128)]:
        rows += check_d_head(net, NB, H, W, N1, N2, seed=NB, bf16=bf16, tol=tol)
    return rows
