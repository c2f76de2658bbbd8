"""CPU oracle for the EdgeGAN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (sysu-imsl/EdgeGAN) ships no tests / golden vectors and its
arithmetic lives in tensorflow_gpu==1.14.0 (requirements.txt:3), which cannot be installed or
run in this environment.  This file is therefore a *restatement* of the reference's algorithm
(file:line cited per function, TF-1.14 op semantics per SURVEY.md Appendix B), written on
torch-CPU so that autograd (incl. create_graph double backward for the WGAN-GP term) supplies
the gradients.  The TF op semantics that torch does not share natively (SAME padding, the
conv2d_transpose alignment, legacy bicubic resize, avg_pool SAME) are additionally pinned by
direct-definition numpy loops in oracle/tf_semantics.py (tests/test_oracle_semantics.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (edgegan_amd/) never imports it and has no CPU fallback.

All image tensors at this API are NHWC (as in the reference's placeholders,
edgegan/models/edgegan.py:185-186); weights use the TF variable layouts (HWIO conv filters,
[kh,kw,Cout,Cin] deconv filters, [in,out] matrices) and the TF variable names of SURVEY.md
Appendix C.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# bf16-operand emulation (for the engine's compute_dtype='bf16' mode; the reference itself is fp32)
# --------------------------------------------------------------------------------------
# Inside `with bf16_operands():` every dense contraction y = f(x, w) (conv, transposed conv, linear) is evaluated
# the way the engine's bf16 mode evaluates it: each of its three bilinear maps -- forward,
# input gradient, filter gradient -- rounds BOTH of its operands to bf16 (round-to-nearest-even) when that map
# qualifies for the bf16 MFMA, and multiplies / accumulates exactly (in the oracle's own precision):
#     y  = f(r(x), r(w))                      if conv rule(K = contracted channels, n_out)
#     dx = f'_x(r(dy), r(w))                  if conv rule with the roles of the channel counts swapped
#     dw = f'_w(r(x), r(dy))                  if wgrad rule
# The backward maps are themselves differentiable (round() is a straight-through estimator there), so the
# WGAN-GP double backward still works; the engine orders that computation differently (reverse-over-forward), so
# for the gradient-penalty sub-steps the emulation is only accurate to bf16 rounding noise, not to fp32 noise.
_BF16_OPERANDS = False


class bf16_operands:
    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        global _BF16_OPERANDS
        self.prev = _BF16_OPERANDS
        _BF16_OPERANDS = bool(self.enabled)

    def __exit__(self, *exc):
        global _BF16_OPERANDS
        _BF16_OPERANDS = self.prev


def bf16_conv_rule(k_channels: int, n_out: int) -> bool:
    return k_channels % 32 == 0 and n_out > 16 and n_out % 4 == 0


def bf16_wgrad_rule(c_gathered: int, n_dense: int) -> bool:
    return c_gathered % 4 == 0 and n_dense % 4 == 0 and c_gathered > 16


def _r16(t: Tensor) -> Tensor:
    d = t.detach()
    return t + (d.to(torch.float32).to(torch.bfloat16).to(t.dtype) - d)


class _RoundedBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, f, rf, rdx, rdw):
        ctx.f, ctx.flags = f, (rdx, rdw)
        ctx.save_for_backward(x, w)
        return f(_r16(x), _r16(w)) if rf else f(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        rdx, rdw = ctx.flags
        f = ctx.f
        dx = dw = None
        with torch.enable_grad():
            if ctx.needs_input_grad[0]:
                xv = x.detach().requires_grad_()
                dx = torch.autograd.grad(f(xv, _r16(w) if rdx else w), xv, _r16(dy) if rdx else dy, create_graph=True)[0]
            if ctx.needs_input_grad[1]:
                wv = w.detach().requires_grad_()
                dw = torch.autograd.grad(f(_r16(x) if rdw else x, wv), wv, _r16(dy) if rdw else dy, create_graph=True)[0]
        return dx, dw, None, None, None, None


def _bilinear(f, x: Tensor, w: Tensor, c_in: int, n_out: int, c_wgrad: Optional[int] = None,
              n_wgrad: Optional[int] = None) -> Tensor:
    """f(x, w), with the engine's bf16 operand rounding when emulation is on.  c_in / n_out: contracted and produced
    channel counts of the forward map; (c_wgrad, n_wgrad): gathered / dense operand channels of the engine's wgrad
    launch (default: the forward's)."""
    if not _BF16_OPERANDS:
        return f(x, w)
    rf = bf16_conv_rule(c_in, n_out)
    rdx = bf16_conv_rule(n_out, c_in)
    rdw = bf16_wgrad_rule(c_in if c_wgrad is None else c_wgrad, n_out if n_wgrad is None else n_wgrad)
    if not (rf or rdx or rdw):
        return f(x, w)
    return _RoundedBilinear.apply(x, w, f, rf, rdx, rdw)



# --------------------------------------------------------------------------------------
# kink recorder (fixture conditioning; tests/golden/select_fixtures.py, tests/test_margin_gpu.py)
# --------------------------------------------------------------------------------------
# ReLU / LeakyReLU / PReLU / |.| have discontinuous derivatives at 0.  An fp32 implementation whose
# pre-activation differs from the fp64 value by rounding noise takes the OTHER slope when the fp64 value lies
# within that noise of 0; in a layer with a small normalisation extent (InstanceNorm over 4x4 .. 16x16) one
# such flip moves that channel's gradient by ~1/(B*H*W).  Inside `with kink_recorder() as rec:` every
# piecewise-linear activation records, per call, how close its argument comes to the kink relative to the
# argument's RMS -- fixtures whose small-extent layers keep a margin can then be compared at tight tolerance.
_KINK_REC = None
_KINK_FLIP = None      # kink_flipper: {activation call index: LongTensor of flat element indices}
_KINK_CALL = 0         # activation calls since the recorder / flipper was entered (the two count alike on the same graph)


class kink_recorder:
    """rec.rows: list of (tag, elements per (n, c) slice, numel, min|x|/rms(x)).
    With collect=(max_extent, margin): rec.near additionally lists (call index, flat element index, |x|/rms, tag, shape) of every non-zero
    argument closer than `margin` to the kink in calls whose per-(n, c) extent is <= max_extent -- the elements an fp32
    evaluation may decide differently (tests/test_tight_gpu.py proves its flip tolerance with them through kink_flipper)."""

    def __init__(self, collect=None):
        self.collect = collect

    def __enter__(self):
        global _KINK_REC, _KINK_CALL
        self.prev, self.rows, self.near = _KINK_REC, [], []
        _KINK_REC = self
        _KINK_CALL = 0
        return self

    def __exit__(self, *exc):
        global _KINK_REC
        _KINK_REC = self.prev

    def margin(self, max_extent: int = 1 << 30) -> float:
        """Smallest recorded margin over the calls whose per-(n,c) extent is <= max_extent."""
        m = [r[3] for r in self.rows if r[1] <= max_extent]
        return min(m) if m else float('inf')


class kink_flipper:
    """Re-evaluate a graph with the listed arguments of its piecewise-linear activations on the OTHER side of the kink (mirrored:
    the elements are within rounding of it): flips = [(call index, flat element index), ...] as recorded by
    kink_recorder(collect=...) on the same graph."""

    def __init__(self, flips):
        self.flips = {}
        for call, idx in flips:
            self.flips.setdefault(int(call), []).append(int(idx))

    def __enter__(self):
        global _KINK_FLIP, _KINK_CALL
        self.prev = _KINK_FLIP
        _KINK_FLIP = self.flips
        _KINK_CALL = 0
        return self

    def __exit__(self, *exc):
        global _KINK_FLIP
        _KINK_FLIP = self.prev


def _flipped(x: Tensor, call: int) -> Tensor:
    """The activation's argument with the elements kink_flipper lists for this call mirrored to the other side of the kink
    (x_i -> -x_i: a change of ~1e-6 of the layer's scale, what an fp32 evaluation's rounding does to such an element).  The mask
    of this activation AND of whatever consumes the value downstream (the encoder's relu -> avg_pool -> relu, whose second relu
    sees an exact zero when all four inputs are negative) follow from the value, as they do in the engine."""
    if _KINK_FLIP is None or call not in _KINK_FLIP:
        return x
    idx = torch.tensor(_KINK_FLIP[call], dtype=torch.long)
    m = torch.zeros(x.numel(), dtype=x.dtype)
    m[idx] = 2.0
    return x - m.reshape(x.shape) * x.detach()


def _kink(tag: str, x: Tensor) -> int:
    """Returns the index of this activation call (for _flipped)."""
    global _KINK_CALL
    call = _KINK_CALL
    _KINK_CALL += 1
    if _KINK_REC is None or x.numel() == 0:
        return call
    d = x.detach()
    if d.dim() == 4:            # NHWC or NCHW: either way numel / (n * channels) elements per slice
        ch = d.shape[3] if tag.endswith('nhwc') else d.shape[1]
        extent = d.numel() // (d.shape[0] * ch)
    else:
        extent = 1
    rms = float(d.pow(2).mean().sqrt())
    nz = d[d != 0]      # exact zeros are structural (relu of an already rectified, pooled tensor), the same in fp32
    _KINK_REC.rows.append((tag, extent, d.numel(), (float(nz.abs().min()) if nz.numel() else float('inf')) / max(rms, 1e-300)))
    if _KINK_REC.collect is not None and extent <= _KINK_REC.collect[0] and rms > 0:
        a = d.reshape(-1).abs()
        for i in torch.nonzero((a < _KINK_REC.collect[1] * rms) & (a != 0)).reshape(-1).tolist():
            _KINK_REC.near.append((call, i, float(a[i]) / rms, tag, tuple(d.shape)))
    return call


def relu(x: Tensor) -> Tensor:
    """tf.nn.relu (slope 0 at 0)."""
    call = _kink('relu/nhwc', x)
    return torch.relu(_flipped(x, call))


# --------------------------------------------------------------------------------------
# configuration (defaults = edgegan/train.py:14-74, edgegan/models/edgegan.py:46-48)
# --------------------------------------------------------------------------------------
@dataclass
class Config:
    batch_size: int = 64
    height: int = 64            # input_height/output_height  train.py:21-26
    width: int = 128            # input_width/output_width    train.py:23-28
    num_classes: int = 14       # 0 => --nomulticlasses       train.py:44-45
    z_dim: int = 100            # train.py:74
    gf_dim: int = 64            # edgegan.py:47
    df_dim: int = 64
    c_dim: int = 3
    dis_size: int = 128         # image_dis_size / edge_dis_size  train.py:64,67 (when not given separately below)
    lambda_gp: float = 10.0     # train.py:52
    zl_weight: float = 10.0     # stage1_zl_loss train.py:41
    learning_rate: float = 2e-4  # train.py:18
    joint_dweight: float = 1.0
    image_dweight: float = 1.0
    edge_dweight: float = 1.0
    use_image_discriminator: bool = True     # train.py:61-62
    use_edge_discriminator: bool = True      # train.py:65-66
    image_dis_size: Optional[int] = None     # train.py:63-64
    edge_dis_size: Optional[int] = None      # train.py:67
    E_norm: Optional[str] = 'instance'       # train.py:54-59: 'instance' | 'batch' | None (normalization.py:10-29)
    G_norm: Optional[str] = 'instance'
    D_norm: Optional[str] = 'instance'

    @property
    def image_dis(self) -> int:
        return self.dis_size if self.image_dis_size is None else self.image_dis_size

    @property
    def edge_dis(self) -> int:
        return self.dis_size if self.edge_dis_size is None else self.edge_dis_size

    @property
    def multiclasses(self) -> bool:
        return self.num_classes > 0

    @property
    def g_in(self) -> int:       # generator input width: z (+ one-hot)  edgegan.py:188-197
        return self.z_dim + (self.num_classes if self.multiclasses else 0)


# --------------------------------------------------------------------------------------
# TF-semantics primitives
# --------------------------------------------------------------------------------------
def _nchw(x: Tensor) -> Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: Tensor) -> Tensor:
    return x.permute(0, 2, 3, 1)


def same_pads(n: int, k: int, s: int) -> Tuple[int, int]:
    """TF SAME padding: out=ceil(n/s); total=max((out-1)s+k-n,0);
    before=total//2."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d(x: Tensor, w: Tensor, stride: int, pad: str, b: Optional[Tensor] = None) -> Tensor:
    """edgegan/nn/modules/conv.py:13-36 (`conv2d`): NHWC input, HWIO filter; pad in
    {'SAME','VALID','REFLECT'}; REFLECT pads p=(k-1)//2 with tf.pad REFLECT then VALID conv
    (conv.py:23-26); optional bias (conv.py:31-34)."""
    k = w.shape[0]
    xc = _nchw(x)
    if pad == 'REFLECT':
        p = (k - 1) // 2
        if p > 0:
            xc = F.pad(xc, (p, p, p, p), mode='reflect')
    elif pad == 'SAME':
        pt, pb = same_pads(x.shape[1], k, stride)
        pl, pr = same_pads(x.shape[2], k, stride)
        xc = F.pad(xc, (pl, pr, pt, pb))
    else:
        assert pad == 'VALID'
    y = _bilinear(lambda xx, ww: F.conv2d(xx, ww.permute(3, 2, 0, 1), stride=stride), xc, w, w.shape[2], w.shape[3])
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


def deconv2d(x: Tensor, w: Tensor, b: Tensor, out_hw: Tuple[int, int]) -> Tensor:
    """edgegan/nn/modules/conv.py:39-58 (`deconv2d`): tf.nn.conv2d_transpose SAME, 5x5 stride 2,
    filter [kh,kw,Cout,Cin], + bias.  TF defines it as the input-gradient of the k5-s2-SAME
    conv: y[o] += x[i] w[k], o = 2i + k - 1, cropped to out_hw."""
    assert w.shape[0] == 5 and w.shape[1] == 5
    # engine wgrad launch for a transposed conv: gathered operand = dy (Cout channels), dense operand = x (Cin)
    y = _bilinear(lambda xx, ww: F.conv_transpose2d(xx, ww.permute(3, 2, 0, 1), stride=2, padding=1)[:, :, :out_hw[0], :out_hw[1]],
                  _nchw(x), w, w.shape[3], w.shape[2], c_wgrad=w.shape[2], n_wgrad=w.shape[3])
    assert y.shape[2] == out_hw[0] and y.shape[3] == out_hw[1]
    y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


def instance_norm(x: Tensor) -> Tensor:
    """edgegan/nn/modules/normalization.py:13-18: tf.nn.moments over H,W (biased variance);
    (x-mean)/(sqrt(var)+1e-5) -- epsilon OUTSIDE the sqrt, no affine (quirk N2)."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) / (torch.sqrt(var) + 1e-5)


def batch_norm_train(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5):
    """normalization.py:19-25: tf.contrib.layers.batch_norm(is_training=True, scale, center,
    epsilon=1e-5, decay=0.9, updates_collections=None) -- batch statistics always (quirk N1);
    biased variance inside the sqrt.  Returns (y, batch_mean, batch_var_biased)."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var


def activation(x: Tensor, name: Optional[str]) -> Tensor:
    """edgegan/nn/modules/activation.py:4-15."""
    if name == 'relu':
        return relu(x)
    if name == 'lrelu':
        x = _flipped(x, _kink('lrelu/nhwc', x))
        return torch.maximum(x, 0.2 * x)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    assert name is None
    return x


def norm(x: Tensor, kind: Optional[str]) -> Tensor:
    if kind == 'instance':
        return instance_norm(x)
    assert kind is None
    return x


def _bn_specs(scope: str, c: int):
    """The variables contrib batch_norm creates under `<scope>/batch_norm` (normalization.py:20-25)."""
    return [VarSpec(f'{scope}/batch_norm/BatchNorm/beta', (c,), True, 'const:0'),
            VarSpec(f'{scope}/batch_norm/BatchNorm/gamma', (c,), True, 'const:1'),
            VarSpec(f'{scope}/batch_norm/BatchNorm/moving_mean', (c,), False, 'const:0'),
            VarSpec(f'{scope}/batch_norm/BatchNorm/moving_variance', (c,), False, 'const:1')]


def block_norm(p, scope: str, x: Tensor, kind: Optional[str]) -> Tensor:
    """`_norm(out, is_train, norm)` inside conv_block / deconv_block / residual (conv.py:65,75,80,128): 'instance', or 'batch'
    = contrib batch_norm over ALL rows of this call with the block's own beta / gamma (normalization.py:19-25; the moving
    statistics never enter a result: is_training=True)."""
    if kind == 'batch':
        y, _, _ = batch_norm_train(x, p[f'{scope}/batch_norm/BatchNorm/gamma'], p[f'{scope}/batch_norm/BatchNorm/beta'])
        return y
    return norm(x, kind)


# Test instrumentation (tests/test_tight_gpu.py): when a dict, every PReLU records sum_i |t_i| of the per-element terms
# t_i = dL/d(leak) contributed by element i (the slope's gradient is sum_i t_i) under id(leak) -- the conditioning
# sum|t| / |sum t| of that scalar sum, which is what a relative tolerance on it has to be scaled by.
_SLOPE_TERMS = None


def prelu(x: Tensor, leak: Tensor) -> Tensor:
    """activation.py:23-27: tf.maximum(leak*x, x), scalar learnable `param` (init 0.2)."""
    x = _flipped(x, _kink('prelu/nchw', x))
    if _SLOPE_TERMS is not None and leak.requires_grad:
        le = leak * torch.ones_like(x)       # same value per element; its gradient holds the terms t_i

        def _rec(g, key=id(leak)):
            _SLOPE_TERMS[key] = _SLOPE_TERMS.get(key, 0.0) + float(g.abs().sum())
        le.register_hook(_rec)
        return torch.maximum(le * x, x)
    return torch.maximum(leak * x, x)


def lrelu_leak(x: Tensor, leak: float = 0.2) -> Tensor:
    """activation.py:30-32."""
    x = _flipped(x, _kink('lrelu_leak/nchw', x))
    return torch.maximum(leak * x, x)


def avg_pool2(x: Tensor) -> Tensor:
    """tf.nn.avg_pool 2x2 s2 SAME on even extents (encoder.py:68)."""
    return _nhwc(F.avg_pool2d(_nchw(x), 2, 2))


def avg_pool_same(x: Tensor, k: int) -> Tensor:
    """tf.nn.avg_pool kxk stride k SAME (encoder.py:70): divides by the number of VALID elements
    (SURVEY Appendix B.5)."""
    xc = _nchw(x)
    H, W = xc.shape[2], xc.shape[3]
    pt, pb = same_pads(H, k, k)
    pl, pr = same_pads(W, k, k)
    xc = F.pad(xc, (pl, pr, pt, pb))
    ones = F.pad(torch.ones(1, 1, H, W, dtype=x.dtype), (pl, pr, pt, pb))
    s = F.avg_pool2d(xc, k, k) * (k * k)
    cnt = F.avg_pool2d(ones, k, k) * (k * k)
    return _nhwc(s / cnt)


def mean_pool_nchw(x: Tensor) -> Tensor:
    """edgegan/nn/modules/pooling.py:4-8 (NCHW 2x2 mean by strided slices)."""
    return (x[:, :, ::2, ::2] + x[:, :, 1::2, ::2] + x[:, :, ::2, 1::2] + x[:, :, 1::2, 1::2]) / 4.0


_BICUBIC_TABLE = None


def _bicubic_table() -> np.ndarray:
    """TF legacy bicubic coefficient table (core/kernels/resize_bicubic_op.cc, kTableSize=1024,
    A=-0.75), float32 like TF."""
    global _BICUBIC_TABLE
    if _BICUBIC_TABLE is None:
        a = -0.75
        tab = np.zeros((1025, 2), dtype=np.float64)
        for i in range(1025):
            x = i / 1024.0
            tab[i, 0] = ((a + 2) * x - (a + 3)) * x * x + 1
            x += 1.0
            tab[i, 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
        _BICUBIC_TABLE = tab.astype(np.float32).astype(np.float64)
    return _BICUBIC_TABLE


def bicubic_matrix(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] interpolation matrix of tf.image.resize_images(method=2) in TF 1.14
    (legacy kernel: align_corners=False, no half-pixel centres)."""
    tab = _bicubic_table()
    R = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        loc = scale * o
        fl = math.floor(loc)
        delta = loc - fl
        off = int(np.rint(delta * 1024))
        wts = (tab[off, 1], tab[off, 0], tab[1024 - off, 0], tab[1024 - off, 1])
        for t, wt in enumerate(wts):
            idx = min(max(fl - 1 + t, 0), n_in - 1)
            R[o, idx] += wt
    return R


def resize_bicubic(x: Tensor, size: int) -> Tensor:
    """edgegan/models/edgegan.py:211-213: tf.image.resize_images(x, [size,size], method=2).
    Identity if the size already matches (TF python wrapper short-circuit)."""
    H, W = x.shape[1], x.shape[2]
    if H == size and W == size:
        return x
    Rh = torch.from_numpy(bicubic_matrix(H, size)).to(x.dtype)
    Rw = torch.from_numpy(bicubic_matrix(W, size)).to(x.dtype)
    y = torch.einsum('oh,nhwc->nowc', Rh, x)
    y = torch.einsum('pw,nowc->nopc', Rw, y)
    return y


def _l2normalize(v: Tensor, eps: float = 1e-12) -> Tensor:
    """normalization.py:34-35."""
    return v / (torch.sum(v ** 2) ** 0.5 + eps)


def spectral_normed_weight(W: Tensor, u: Tensor) -> Tensor:
    """normalization.py:38-76 with num_iters=1 and an update collection: one power iteration from
    the (never-updated, quirk N5) variable u; sigma = v W u'^T; W/sigma; everything differentiable
    w.r.t. W (no stop_gradient)."""
    shape = W.shape
    Wr = W.reshape(-1, shape[-1])
    v = _l2normalize(u @ Wr.t())
    u1 = _l2normalize(v @ Wr)
    sigma = (v @ Wr @ u1.t())[0, 0]
    return (Wr / sigma).reshape(shape)


def linear(x: Tensor, W: Tensor, b: Tensor) -> Tensor:
    """edgegan/nn/modules/linear.py:10-31 / mlp :79-92."""
    return _bilinear(lambda xx, ww: xx @ ww, x, W, W.shape[0], W.shape[1]) + b


# --------------------------------------------------------------------------------------
# variable specification (names/shapes = SURVEY Appendix C; initialisers = Appendix B.9)
# --------------------------------------------------------------------------------------
@dataclass
class VarSpec:
    name: str
    shape: Tuple[int, ...]
    trainable: bool
    init: str              # 'normal:0.02' | 'trunc:0.02' | 'trunc:1.0' | 'const:V' | 'xavier_uniform'


def _ceil_half(n: int) -> int:
    return -(-n // 2)


def generator_specs(prefix: str, cfg: Config) -> List[VarSpec]:
    """edgegan/models/generator.py:35-74 (_convnet), output (H, W/2)."""
    h, w = cfg.height, cfg.width // 2
    s16 = (h, w)
    for _ in range(4):
        s16 = (_ceil_half(s16[0]), _ceil_half(s16[1]))
    gf = cfg.gf_dim
    v = [
        VarSpec(f'{prefix}/g_lin_0/Matrix', (cfg.g_in, gf * 8 * s16[0] * s16[1]), True, 'normal:0.02'),
        VarSpec(f'{prefix}/g_lin_0/bias', (gf * 8 * s16[0] * s16[1],), True, 'const:0'),
        VarSpec(f'{prefix}/batch_norm/BatchNorm/beta', (gf * 8,), True, 'const:0'),
        VarSpec(f'{prefix}/batch_norm/BatchNorm/gamma', (gf * 8,), True, 'const:1'),
        VarSpec(f'{prefix}/batch_norm/BatchNorm/moving_mean', (gf * 8,), False, 'const:0'),
        VarSpec(f'{prefix}/batch_norm/BatchNorm/moving_variance', (gf * 8,), False, 'const:1'),
    ]
    chans = [gf * 8, gf * 4, gf * 2, gf, cfg.c_dim]
    for i in range(1, 5):
        v.append(VarSpec(f'{prefix}/g_dconv_{i}/deconv2d/w', (5, 5, chans[i], chans[i - 1]), True, 'normal:0.02'))
        v.append(VarSpec(f'{prefix}/g_dconv_{i}/deconv2d/b', (chans[i],), True, 'const:0'))
        if cfg.G_norm == 'batch' and i < 4:
            v += _bn_specs(f'{prefix}/g_dconv_{i}', chans[i])
    return v


def discriminator_specs(prefix: str, cfg: Config, in_h: int, in_w: int) -> List[VarSpec]:
    """edgegan/models/discriminator.py:58-81 (_convnet)."""
    df = cfg.df_dim
    chans = [cfg.c_dim, df, df * 2, df * 4, df * 8]
    names = ['d_conv_0', 'd_conv_1', 'd_conv_3', 'd_conv_4']
    v = []
    h, w = in_h, in_w
    for i, n in enumerate(names):
        v.append(VarSpec(f'{prefix}/{n}/conv2d/w', (4, 4, chans[i], chans[i + 1]), True, 'trunc:0.02'))
        if cfg.D_norm == 'batch' and i > 0:
            v += _bn_specs(f'{prefix}/{n}', chans[i + 1])
        h, w = _ceil_half(h), _ceil_half(w)
    v.append(VarSpec(f'{prefix}/d_linear_5/Matrix', (h * w * chans[-1], 1), True, 'normal:0.02'))
    v.append(VarSpec(f'{prefix}/d_linear_5/bias', (1,), True, 'const:0'))
    return v


def encoder_filters(cfg: Config) -> List[int]:
    nf = [128, 256, 512, 512]
    if cfg.height == 256:           # encoder.py:57-58
        nf.append(512)
    return nf


def encoder_specs(cfg: Config) -> List[VarSpec]:
    """edgegan/models/encoder.py:54-84 (_resnet), conv.py:70-85 (residual, bias=True)."""
    v = [VarSpec('E/e_resnet_64_0/conv2d/w', (4, 4, cfg.c_dim, 64), True, 'trunc:0.02'),
         VarSpec('E/e_resnet_64_0/conv2d/b', (64,), True, 'const:0')]
    cin = 64
    for i, n in enumerate(encoder_filters(cfg)):
        base = f'E/e_resnet_{n}_{i + 1}'
        bn = cfg.E_norm == 'batch'
        v += [VarSpec(f'{base}/res1/conv2d/w', (3, 3, cin, n), True, 'trunc:0.02'),
              VarSpec(f'{base}/res1/conv2d/b', (n,), True, 'const:0')] + (_bn_specs(f'{base}/res1', n) if bn else []) + [
              VarSpec(f'{base}/res2/conv2d/w', (3, 3, n, n), True, 'trunc:0.02'),
              VarSpec(f'{base}/res2/conv2d/b', (n,), True, 'const:0')] + (_bn_specs(f'{base}/res2', n) if bn else []) + [
              VarSpec(f'{base}/shortcut/conv2d/w', (1, 1, cin, n), True, 'trunc:0.02'),
              VarSpec(f'{base}/shortcut/conv2d/b', (n,), True, 'const:0')]
        cin = n
    for head in ('FC8_mu', 'FC8_sigma'):
        v += [VarSpec(f'E/{head}/w', (cin, cfg.z_dim), True, 'normal:0.02'),
              VarSpec(f'E/{head}/b', (cfg.z_dim,), True, 'const:0')]
    return v


def _sn_u_name(scope: str) -> str:
    # normalization.py:42-44: tf.variable_scope(W.name.rsplit('/',1)[0]) entered INSIDE the current
    # scope => nested name (unverifiable without TF).
    return f'{scope}/{scope}/u'


def classifier_specs(cfg: Config) -> List[VarSpec]:
    """edgegan/models/classifier.py:12-119 + conv.py:133-357 (mru_conv / mru_conv_block_v3 /
    conv2d2), linear.py:34-76 (fully_connected)."""
    v: List[VarSpec] = []

    def conv(scope, k, cin, cout, bias_init='const:0', prelu_param=False):
        v.append(VarSpec(f'{scope}/weights', (k, k, cin, cout), True, 'normal:0.02'))
        v.append(VarSpec(_sn_u_name(scope), (1, cout), False, 'trunc:1.0'))
        v.append(VarSpec(f'{scope}/biases', (1, cout, 1, 1), True, bias_init))
        if prelu_param:
            v.append(VarSpec(f'{scope}/prelu/param', (), True, 'const:0.2'))

    conv('D2/Conv', 7, 3, 8, prelu_param=True)
    hidden = 8
    for unit, depth in enumerate([128, 256, 512, 768], start=1):
        base = f'D2/mru_conv_unit_t_{unit}_layer_0'
        v.append(VarSpec(f'{base}/norm_activation_in/prelu/param', (), True, 'const:0.2'))
        conv(f'{base}/update_gate', 3, hidden + 3, hidden, bias_init='const:0.5')
        conv(f'{base}/Conv', 3, 3, hidden)
        v.append(VarSpec(f'{base}/norm_activation_merge_1/prelu/param', (), True, 'const:0.2'))
        conv(f'{base}/Conv_1', 3, hidden, depth, prelu_param=True)
        conv(f'{base}/Conv_2', 3, depth, depth)
        conv(f'{base}/Conv_3', 1, hidden, depth)
        hidden = depth
    v.append(VarSpec('D2/mru_conv_unit_last_norm/prelu/param', (), True, 'const:0.2'))
    conv('D2/Conv_1', 1, hidden, 1)          # dead `disc` head (quirk N8)
    v.append(VarSpec('D2/fully_connected/weights', (hidden, cfg.num_classes), True, 'xavier_uniform'))
    v.append(VarSpec(_sn_u_name('D2/fully_connected'), (1, cfg.num_classes), False, 'trunc:1.0'))
    v.append(VarSpec('D2/fully_connected/biases', (cfg.num_classes,), True, 'const:0'))
    return v


def variable_specs(cfg: Config) -> List[VarSpec]:
    """Creation order of edgegan/models/edgegan.py:202-275 (forward): G1, G2, D2, D, D_patch2,
    D_patch3, E."""
    v = generator_specs('G1', cfg) + generator_specs('G2', cfg)
    if cfg.multiclasses:
        v += classifier_specs(cfg)
    v += discriminator_specs('D', cfg, cfg.height, cfg.width)
    if cfg.use_image_discriminator:          # edgegan.py:164-168: not constructed when switched off
        v += discriminator_specs('D_patch2', cfg, cfg.image_dis, cfg.image_dis)
    if cfg.use_edge_discriminator:           # edgegan.py:170-174
        v += discriminator_specs('D_patch3', cfg, cfg.edge_dis, cfg.edge_dis)
    v += encoder_specs(cfg)
    return v


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
P = Dict[str, Tensor]


def generator(p: P, prefix: str, z: Tensor, cfg: Config, bn_stats: Optional[list] = None) -> Tensor:
    """edgegan/models/generator.py:35-74.  First norm is BatchNorm (quirk N1, generator.py:51-52);
    deconv blocks 1-3 IN+ReLU, block 4 no norm/act, then tanh (generator.py:74)."""
    gf = cfg.gf_dim
    h, w = cfg.height, cfg.width // 2
    sizes = [(h, w)]
    for _ in range(4):
        sizes.append((_ceil_half(sizes[-1][0]), _ceil_half(sizes[-1][1])))
    z_ = linear(z, p[f'{prefix}/g_lin_0/Matrix'], p[f'{prefix}/g_lin_0/bias'])
    h0 = z_.reshape(-1, sizes[4][0], sizes[4][1], gf * 8)
    h0, bm, bv = batch_norm_train(h0, p[f'{prefix}/batch_norm/BatchNorm/gamma'],
                                  p[f'{prefix}/batch_norm/BatchNorm/beta'])
    if bn_stats is not None:
        bn_stats.append((bm.detach(), bv.detach(), h0.shape[0] * h0.shape[1] * h0.shape[2]))
    x = relu(h0)
    for i in range(1, 5):
        x = deconv2d(x, p[f'{prefix}/g_dconv_{i}/deconv2d/w'], p[f'{prefix}/g_dconv_{i}/deconv2d/b'],
                     sizes[4 - i])
        if i < 4:
            x = relu(block_norm(p, f'{prefix}/g_dconv_{i}', x, cfg.G_norm))
    return torch.tanh(x)


def discriminator(p: P, prefix: str, x: Tensor, norm_kind: str = 'instance') -> Tuple[Tensor, Tensor]:
    """edgegan/models/discriminator.py:58-81: returns (sigmoid(D), D).  norm_kind = D_norm (train.py:58-59)."""
    names = ['d_conv_0', 'd_conv_1', 'd_conv_3', 'd_conv_4']
    for i, n in enumerate(names):
        x = conv2d(x, p[f'{prefix}/{n}/conv2d/w'], 2, 'SAME')
        if i > 0:
            x = block_norm(p, f'{prefix}/{n}', x, norm_kind)
        x = activation(x, 'lrelu')
    x = x.reshape(x.shape[0], -1)
    d = linear(x, p[f'{prefix}/d_linear_5/Matrix'], p[f'{prefix}/d_linear_5/bias'])
    return torch.sigmoid(d), d


def residual(p: P, base: str, x: Tensor, norm_kind: str = 'instance') -> Tensor:
    """edgegan/nn/modules/conv.py:70-85 with norm = E_norm ('instance' by default), pad='REFLECT', bias=True."""
    out = conv2d(x, p[f'{base}/res1/conv2d/w'], 1, 'REFLECT', p[f'{base}/res1/conv2d/b'])
    out = relu(block_norm(p, f'{base}/res1', out, norm_kind))
    out = conv2d(out, p[f'{base}/res2/conv2d/w'], 1, 'REFLECT', p[f'{base}/res2/conv2d/b'])
    out = block_norm(p, f'{base}/res2', out, norm_kind)
    sc = conv2d(x, p[f'{base}/shortcut/conv2d/w'], 1, 'REFLECT', p[f'{base}/shortcut/conv2d/b'])
    return relu(sc + out)


def encoder(p: P, x: Tensor, eps, cfg: Config) -> Tuple[Tensor, Tensor, Tensor]:
    """edgegan/models/encoder.py:54-84.  eps is ONE scalar N(0,1) draw per run (quirk N3,
    encoder.py:78-79), supplied explicitly."""
    E = conv2d(x, p['E/e_resnet_64_0/conv2d/w'], 2, 'SAME', p['E/e_resnet_64_0/conv2d/b'])
    E = relu(E)
    for i, n in enumerate(encoder_filters(cfg)):
        E = residual(p, f'E/e_resnet_{n}_{i + 1}', E, cfg.E_norm)
        E = avg_pool2(E)
    E = relu(E)
    E = avg_pool_same(E, 8)
    E = E.reshape(E.shape[0], -1)
    mu = linear(E, p['E/FC8_mu/w'], p['E/FC8_mu/b'])
    log_sigma = linear(E, p['E/FC8_sigma/w'], p['E/FC8_sigma/b'])
    z = mu + eps * torch.exp(log_sigma)
    return z, mu, log_sigma


def _sn_conv(p: P, scope: str, x: Tensor, k: int, act=None, split: int = 0) -> Tensor:
    """conv2d2 (conv.py:246-295): NCHW, SAME stride 1, SN weights, + biases (1,C,1,1), optional act.
    `split` only matters under bf16_operands(): the engine evaluates the gate conv over concat([h, x]) as
    conv(h, W[:, :, :split]) + conv(x, W[:, :, split:]), and the bf16 rule applies to each part separately."""
    W = spectral_normed_weight(p[f'{scope}/weights'], p[_sn_u_name(scope)])
    pd = (k - 1) // 2
    conv = lambda xx, ww: F.conv2d(xx, ww.permute(3, 2, 0, 1), padding=pd)
    if _BF16_OPERANDS and split:
        y = _bilinear(conv, x[:, :split], W[:, :, :split], split, W.shape[3]) + \
            _bilinear(conv, x[:, split:], W[:, :, split:], W.shape[2] - split, W.shape[3])
    else:
        y = _bilinear(conv, x, W, W.shape[2], W.shape[3])
    y = y + p[f'{scope}/biases']
    if act is not None:
        y = act(y)
    return y


def mru_block(p: P, base: str, inp: Tensor, ht: Tensor, depth: int) -> Tensor:
    """mru_conv_block_v3 (conv.py:133-243) as driven by mru_conv (conv.py:298-357) with
    num_blocks=1, stride=2, activation prelu, no normaliser, NCHW."""
    hidden = ht.shape[1]
    ht_orig = ht
    full = torch.cat([prelu(ht, p[f'{base}/norm_activation_in/prelu/param']), inp], dim=1)
    rg = _sn_conv(p, f'{base}/update_gate', full, 3, act=lrelu_leak, split=hidden)
    mn = rg.amin(dim=(2, 3), keepdim=True)
    mx = rg.amax(dim=(2, 3), keepdim=True)
    rg = (rg - mn) / (mx - mn)
    img_new = _sn_conv(p, f'{base}/Conv', inp, 3)
    ht_plus = ht + rg * img_new
    ht_new_in = prelu(ht_plus, p[f'{base}/norm_activation_merge_1/prelu/param'])
    h_new = _sn_conv(p, f'{base}/Conv_1', ht_new_in, 3,
                     act=lambda t: prelu(t, p[f'{base}/Conv_1/prelu/param']))
    h_new = _sn_conv(p, f'{base}/Conv_2', h_new, 3)
    if hidden != depth:
        ht_orig = _sn_conv(p, f'{base}/Conv_3', ht_orig, 1)
    return mean_pool_nchw(ht_orig + h_new)


def classifier(p: P, x_nhwc: Tensor, cfg: Config) -> Tensor:
    """edgegan/models/classifier.py:12-119 on channel_first(x) (edgegan.py:28-29,229-236);
    returns the logits (third return value; `disc` head is dead, quirk N8)."""
    x = _nchw(x_nhwc)
    x_list = [x]
    for _ in range(5):
        x_list.append(mean_pool_nchw(x_list[-1]))
    x_list = x_list[::-1]
    h = _sn_conv(p, 'D2/Conv', x_list[-1], 7, act=lambda t: prelu(t, p['D2/Conv/prelu/param']))
    for unit, depth in enumerate([128, 256, 512, 768], start=1):
        h = mru_block(p, f'D2/mru_conv_unit_t_{unit}_layer_0', x_list[-unit], h, depth)
    h = prelu(h, p['D2/mru_conv_unit_last_norm/prelu/param'])      # conv.py:353-355 (last_unit)
    img = h.mean(dim=(2, 3))
    W = spectral_normed_weight(p['D2/fully_connected/weights'], p[_sn_u_name('D2/fully_connected')])
    return img @ W + p['D2/fully_connected/biases']


# --------------------------------------------------------------------------------------
# losses (edgegan/nn/functional.py)
# --------------------------------------------------------------------------------------
def discriminator_ganloss(output: Tensor, target: Tensor) -> Tensor:
    return torch.mean(output - target)                              # functional.py:32-33


def generator_ganloss(output: Tensor) -> Tensor:
    return torch.mean(output * -1)                                  # functional.py:36-37


def l1loss(output: Tensor, target: Tensor, weight: float) -> Tensor:
    _kink('abs', output - target)
    return weight * torch.mean(torch.abs(output - target))         # functional.py:40-41


def acgan_loss_focal(real_logits: Tensor, labels: Tensor, fake_logits: Tensor,
                     ld1=1.0, ld2=0.5, ld_focal=2.0) -> Tuple[Tensor, Tensor]:
    """functional.py:5-16 -> (loss_ac_g, loss_ac_d).  The focal weight is differentiated (no
    stop_gradient)."""
    sm = torch.softmax(real_logits, dim=1)
    py = sm.gather(1, labels.view(-1, 1)).squeeze(1)
    ce_r = F.cross_entropy(real_logits, labels, reduction='none')
    loss_d = ld1 * torch.mean((1 - py) ** ld_focal * ce_r)
    loss_g = ld2 * torch.mean(F.cross_entropy(fake_logits, labels, reduction='none'))
    return loss_g, loss_d


def gradient_penalty_term(p: P, prefix: str, synthesized: Tensor, real: Tensor, alpha: Tensor,
                          weight: float, norm_kind: str = 'instance') -> Tensor:
    """edgegan/models/edgegan.py:32-42 (`random_blend`, `penalty`) + functional.py:26-29.

    Quirk N10 (found while restating; not in SURVEY): `penalty` passes the discriminator's whole
    return value -- the TUPLE (sigmoid(D), D) (discriminator.py:81) -- to tf.gradients, which sums
    the gradients of every element of `ys`.  The penalised gradient is therefore
    d/dx [ sum_b sigmoid(D(x_b)) + D(x_b) ], not d/dx sum_b D(x_b)."""
    a = alpha.view(-1, 1, 1, 1)
    inter = real + a * (synthesized - real)                         # b + alpha*(a-b), edgegan.py:35
    if not inter.requires_grad:
        inter = inter.detach().requires_grad_(True)
    sig, logit = discriminator(p, prefix, inter, norm_kind)
    g, = torch.autograd.grad(sig.sum() + logit.sum(), inter, create_graph=True)
    grad_l2 = torch.sqrt(torch.sum(g ** 2, dim=(1, 2, 3)))
    return weight * torch.mean((grad_l2 - 1) ** 2)


# --------------------------------------------------------------------------------------
# the training iteration (edgegan/models/edgegan.py:87-130, 202-342)
# --------------------------------------------------------------------------------------
def z_onehot(z: Tensor, cfg: Config) -> Tuple[Tensor, Optional[Tensor]]:
    """edgegan.py:188-197: last column of z is the class id."""
    if not cfg.multiclasses:
        return z, None
    labels = z[:, -1].to(torch.int64)
    oh = F.one_hot(labels, cfg.num_classes).to(z.dtype)
    return torch.cat([z[:, :cfg.z_dim], oh], dim=1), labels


@dataclass
class StepInputs:
    images: Tensor     # [B,H,W,3] in [-1,1]
    z: Tensor          # [B, z_dim (+1)]
    alpha: Tensor      # [3,B] U(0,1): rows = joint D, image D (patch2), edge D (patch3)
    eps: Tensor        # [>=1] N(0,1): eps[0] used by the e_optim run


SUBSTEPS = ['d_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim', 'g_optim_b']


def _vars_with_prefix(p: P, prefix: str, trainable: Dict[str, bool]) -> List[str]:
    return [n for n in p if n.startswith(prefix + '/') and trainable[n]]


ALL_LOSSES = ('joint_dis_dloss', 'image_dis_dloss', 'edge_dis_dloss', 'loss_d_ac', 'edge_gloss', 'image_gloss',
              'zl_loss')


def forward_graph(p: P, inp: StepInputs, cfg: Config, eps_val, bn_stats=None, want=None) -> Dict[str, Tensor]:
    """edgegan.py:202-275 (forward) + :277-342 (define_losses).  Like sess.run, only the sub-graph that
    the requested loss tensors (`want`, default: all) depend on is evaluated."""
    want = set(want or ALL_LOSSES)
    W2 = cfg.width // 2
    zin, labels = z_onehot(inp.z, cfg)
    real = inp.images
    real_edge, real_pic = real[:, :, :W2, :], real[:, :, W2:, :]
    memo: Dict[str, object] = {}
    out: Dict[str, Tensor] = {}

    def once(key, fn):
        if key not in memo:
            memo[key] = fn()
        return memo[key]

    def gen(name):
        def make():
            st = [] if bn_stats is not None else None
            o = generator(p, name, zin, cfg, st)
            if bn_stats is not None:
                bn_stats[name] = st[0]
            return o
        return once('gen' + name, make)

    edge = lambda: gen('G1')
    image = lambda: gen('G2')
    joint = lambda: once('joint', lambda: torch.cat([edge(), image()], dim=2))
    d_fake = lambda: once('d_fake', lambda: discriminator(p, 'D', joint(), cfg.D_norm)[1])
    r_pic = lambda: once('r_pic', lambda: resize_bicubic(real_pic, cfg.image_dis))
    r_img = lambda: once('r_img', lambda: resize_bicubic(image(), cfg.image_dis))
    di_fake = lambda: once('di_fake', lambda: discriminator(p, 'D_patch2', r_img(), cfg.D_norm)[1])
    r_edges = lambda: once('r_edges', lambda: resize_bicubic(real_edge, cfg.edge_dis))
    r_edge_out = lambda: once('r_edge_out', lambda: resize_bicubic(edge(), cfg.edge_dis))
    de_fake = lambda: once('de_fake', lambda: discriminator(p, 'D_patch3', r_edge_out(), cfg.D_norm)[1])
    joint_g = lambda: once('joint_g', lambda: generator_ganloss(d_fake()))

    if 'joint_dis_dloss' in want:
        _, d_real = discriminator(p, 'D', real, cfg.D_norm)
        out['joint_dis_dloss'] = discriminator_ganloss(d_fake(), d_real) + gradient_penalty_term(
            p, 'D', joint(), real, inp.alpha[0], cfg.lambda_gp, cfg.D_norm)
    if 'image_dis_dloss' in want and cfg.use_image_discriminator:    # edgegan.py:288-301 (else: the constant 0)
        _, di_real = discriminator(p, 'D_patch2', r_pic(), cfg.D_norm)
        out['image_dis_dloss'] = discriminator_ganloss(di_fake(), di_real) + gradient_penalty_term(
            p, 'D_patch2', r_img(), r_pic(), inp.alpha[1], cfg.lambda_gp, cfg.D_norm)
    if 'edge_dis_dloss' in want and cfg.use_edge_discriminator:      # edgegan.py:303-311
        _, de_real = discriminator(p, 'D_patch3', r_edges(), cfg.D_norm)
        out['edge_dis_dloss'] = discriminator_ganloss(de_fake(), de_real) + gradient_penalty_term(
            p, 'D_patch3', r_edge_out(), r_edges(), inp.alpha[2], cfg.lambda_gp, cfg.D_norm)
    if 'edge_gloss' in want:
        out['joint_dis_gloss'] = joint_g()
        out['edge_dis_gloss'] = generator_ganloss(de_fake()) if cfg.use_edge_discriminator else torch.zeros((), dtype=real.dtype)
        out['edge_gloss'] = cfg.joint_dweight * joint_g() + cfg.edge_dweight * out['edge_dis_gloss']
    if 'image_gloss' in want:
        out['joint_dis_gloss'] = joint_g()
        out['image_dis_gloss'] = generator_ganloss(di_fake()) if cfg.use_image_discriminator else torch.zeros((), dtype=real.dtype)
        out['image_gloss'] = cfg.joint_dweight * joint_g() + cfg.image_dweight * out['image_dis_gloss']
    if cfg.multiclasses and ('loss_d_ac' in want or 'image_gloss' in want):
        lr_ = classifier(p, real_pic, cfg) if 'loss_d_ac' in want else None
        lf_ = classifier(p, image(), cfg) if 'image_gloss' in want else None
        # functional.py:5-16 computes both terms in one call; each needs only its own logits
        if lr_ is not None:
            out['loss_d_ac'] = acgan_loss_focal(lr_, labels, lr_.detach())[1]
        if lf_ is not None:
            out['loss_g_ac'] = acgan_loss_focal(lf_.detach(), labels, lf_)[0]
            out['image_gloss'] = out['image_gloss'] + out['loss_g_ac']
    if 'zl_loss' in want:
        z_recon, _, _ = encoder(p, edge(), eps_val, cfg)
        out['zl_loss'] = l1loss(inp.z[:, :cfg.z_dim], z_recon, cfg.zl_weight)
        out['z_recon'] = z_recon
    return out


def rmsprop_apply(w: Tensor, g: Tensor, ms: Tensor, mom: Tensor, lr: float):
    """tf.train.RMSPropOptimizer(lr): decay 0.9, momentum 0, epsilon 1e-10, not centered
    (SURVEY Appendix B.7).  In-place on w, ms, mom."""
    ms.add_((g * g - ms) * (1.0 - 0.9))
    mom.copy_(0.0 * mom + lr * g / torch.sqrt(ms + 1e-10))
    w.sub_(mom)


class OracleModel:
    """Holds variables + RMSProp slots and runs the 7 sub-steps (edgegan.py:109-130)."""

    def __init__(self, cfg: Config, values: Dict[str, np.ndarray], dtype=torch.float64):
        self.cfg = cfg
        self.dtype = dtype
        self.specs = variable_specs(cfg)
        self.trainable = {s.name: s.trainable for s in self.specs}
        self.p: P = {}
        for s in self.specs:
            a = np.asarray(values[s.name]).reshape(s.shape)
            self.p[s.name] = torch.tensor(a, dtype=dtype)
        self.ms = {n: torch.ones_like(t) for n, t in self.p.items() if self.trainable[n]}
        self.mom = {n: torch.zeros_like(t) for n, t in self.p.items() if self.trainable[n]}

    # which variables a sub-step's optimizer(s) touch: (loss key, variable prefix)
    def _substep_targets(self, name: str):
        t = {
            'd_optim': [('joint_dis_dloss', 'D')],
            'd_optim_patch2': [('image_dis_dloss', 'D_patch2')],
            'd_optim_patch3': [('edge_dis_dloss', 'D_patch3')],
            'd_optim2': [('loss_d_ac', 'D2')],
            'g_optim_u': [('edge_gloss', 'G1'), ('image_gloss', 'G2')],
            'e_optim': [('zl_loss', 'E')],
            'g_optim_b': [('edge_gloss', 'G1'), ('image_gloss', 'G2')],
        }[name]
        return t

    def substep_names(self) -> List[str]:
        # register_optim_if (edgegan.py:109-124): d_optim_patch2 / d_optim_patch3 / d_optim2 only with their flag
        off = {'d_optim2': not self.cfg.multiclasses, 'd_optim_patch2': not self.cfg.use_image_discriminator,
               'd_optim_patch3': not self.cfg.use_edge_discriminator}
        return [n for n in SUBSTEPS if not off.get(n, False)]

    def _leaf_params(self) -> P:
        return {n: (t.detach().clone().requires_grad_(True) if self.trainable[n] else t)
                for n, t in self.p.items()}

    def compute_grads(self, name: str, inp: StepInputs, eps_val) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
        """Gradients the sub-step's minimize ops apply (both G ops of g_optim_* are evaluated at the
        pre-update weights of the run, quirk N9).  Returns (grads by var name, losses)."""
        p = self._leaf_params()
        bn = {}
        want = [lk for lk, _ in self._substep_targets(name)]
        losses = forward_graph(p, inp, self.cfg, eps_val, bn_stats=bn, want=want)
        grads: Dict[str, Tensor] = {}
        for loss_key, prefix in self._substep_targets(name):
            names = _vars_with_prefix(p, prefix, self.trainable)
            gs = torch.autograd.grad(losses[loss_key], [p[n] for n in names], retain_graph=True,
                                     allow_unused=True)
            for n, g in zip(names, gs):
                if g is not None:                    # None => variable skipped by minimize (N4, N8)
                    grads[n] = g.detach()
        if _SLOPE_TERMS is not None:                 # (instrumentation) slope name -> sum_i |t_i|
            self.last_slope_l1 = {n: _SLOPE_TERMS[id(t)] for n, t in p.items() if id(t) in _SLOPE_TERMS}
        self._last_bn = bn
        return grads, {k: v.detach() for k, v in losses.items() if v.dim() == 0}

    def apply_grads(self, grads: Dict[str, Tensor]):
        for n, g in grads.items():
            rmsprop_apply(self.p[n], g, self.ms[n], self.mom[n], self.cfg.learning_rate)

    def update_bn_moving(self, which: List[str]):
        """contrib batch_norm with updates_collections=None updates the moving statistics in place
        whenever that generator's forward is evaluated (decay 0.9; moving_variance takes the
        UNBIASED batch variance -- fused batch norm)."""
        for g in which:
            bm, bv, n = self._last_bn[g]
            mm = self.p[f'{g}/batch_norm/BatchNorm/moving_mean']
            mv = self.p[f'{g}/batch_norm/BatchNorm/moving_variance']
            mm.mul_(0.9).add_(0.1 * bm)
            mv.mul_(0.9).add_(0.1 * bv * (n / max(n - 1, 1)))

    # which generators each sub-step's fetched ops depend on
    _GENS = {'d_optim': ['G1', 'G2'], 'd_optim_patch2': ['G2'], 'd_optim_patch3': ['G1'], 'd_optim2': [],
             'g_optim_u': ['G1', 'G2'], 'e_optim': ['G1'], 'g_optim_b': ['G1', 'G2']}

    def run_substep(self, name: str, inp: StepInputs) -> Dict[str, Tensor]:
        eps_val = inp.eps[0]
        grads, losses = self.compute_grads(name, inp, eps_val)
        self.apply_grads(grads)
        self.update_bn_moving(self._GENS[name])
        return losses

    def update_model(self, inp: StepInputs) -> Dict[str, Dict[str, Tensor]]:
        """edgegan.py:126-130: the registered sub-steps in order, same feed each time."""
        return {n: self.run_substep(n, inp) for n in self.substep_names()}

    def infer(self, images: Tensor, classes: Optional[Tensor], eps_edge, eps_image):
        """edgegan.py:492-517,606-609: E(left half) -> z (+one-hot) -> G1 / G2; two separate runs
        => two eps draws."""
        cfg = self.cfg
        left = images[:, :cfg.height, :cfg.width // 2, :cfg.c_dim]
        outs = []
        for eps_val, g in ((eps_edge, 'G1'), (eps_image, 'G2')):
            z, _, _ = encoder(self.p, left, eps_val, cfg)
            if cfg.multiclasses:
                z = torch.cat([z, F.one_hot(classes.to(torch.int64), cfg.num_classes).to(z.dtype)], dim=1)
            outs.append(generator(self.p, g, z, cfg))
        return outs[0], outs[1]

    def numpy_params(self) -> Dict[str, np.ndarray]:
        return {n: t.detach().numpy().copy() for n, t in self.p.items()}
