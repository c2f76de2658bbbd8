"""Pins the oracle's torch formulations of TF-1.14 op semantics against direct-definition numpy loops and
known-answer values.  PARITY UNPINNED by the reference itself (no TF, no goldens)."""
import numpy as np
import pytest
import torch

import edgegan_oracle as O
import tf_semantics as S

RNG = np.random.RandomState(0)


def t(a):
    return torch.tensor(a, dtype=torch.float64)


def test_same_padding_known_answers():
    # Appendix B.1: k4 s2 even -> (1,1); k3 s1 -> (1,1); k7 s1 -> (3,3); k5 s2 even -> (1,2)
    assert O.same_pads(64, 4, 2) == (1, 1) and O.same_pads(64, 3, 1) == (1, 1)
    assert O.same_pads(64, 7, 1) == (3, 3) and O.same_pads(64, 5, 2) == (1, 2)
    assert O.same_pads(2, 8, 8) == (3, 3)
    assert S.same_pad(64, 4, 2) == (32, 1) and S.same_pad(7, 4, 2) == (4, 1)


@pytest.mark.parametrize('k,s,H,W', [(4, 2, 8, 12), (3, 1, 5, 6), (7, 1, 6, 6), (1, 1, 4, 4), (4, 2, 7, 9)])
def test_conv2d_same(k, s, H, W):
    x, w = RNG.normal(size=(2, H, W, 3)), RNG.normal(size=(k, k, 3, 5))
    assert np.allclose(O.conv2d(t(x), t(w), s, 'SAME').numpy(), S.conv2d_same(x, w, s), atol=1e-12)


def test_conv2d_reflect():
    x, w = RNG.normal(size=(2, 5, 6, 3)), RNG.normal(size=(3, 3, 3, 4))
    assert np.allclose(O.conv2d(t(x), t(w), 1, 'REFLECT').numpy(), S.conv2d_reflect(x, w), atol=1e-12)
    # REFLECT excludes the border: padding [a b c] by 1 gives [b a b c b]
    row = torch.tensor([[1., 2., 3.]]).view(1, 1, 3)
    assert torch.nn.functional.pad(row, (1, 1), mode='reflect').flatten().tolist() == [2, 1, 2, 3, 2]


def test_conv2d_transpose_alignment():
    x, w = RNG.normal(size=(2, 3, 4, 6)), RNG.normal(size=(5, 5, 4, 6))
    b = RNG.normal(size=(4,))
    ref = S.conv2d_transpose_same_k5s2(x, w) + b
    assert np.allclose(O.deconv2d(t(x), t(w), t(b), (6, 8)).numpy(), ref, atol=1e-12)
    # and it IS the input-gradient of the forward k5-s2-SAME conv (TF's definition of conv2d_transpose)
    xi = t(RNG.normal(size=(1, 6, 8, 4))).requires_grad_(True)
    wf = t(np.transpose(w, (0, 1, 2, 3)))            # forward filter HWIO with I=Cout(4), O=Cin(6)
    y = O.conv2d(xi, wf, 2, 'SAME')
    g, = torch.autograd.grad((y * t(x[:1])).sum(), xi)
    assert np.allclose(g.numpy(), S.conv2d_transpose_same_k5s2(x[:1], w), atol=1e-12)


def test_bicubic_known_answers_and_definition():
    R = O.bicubic_matrix(64, 128)
    # even outputs copy the source pixel; odd outputs use (-3/32, 19/32, 19/32, -3/32)  (Appendix B.3)
    assert R[10, 5] == 1.0 and np.count_nonzero(R[10]) == 1
    assert np.allclose(R[11, 4:8], [-0.09375, 0.59375, 0.59375, -0.09375]) and np.count_nonzero(R[11]) == 4
    # clamped borders: output 1 touches indices clamp(-1..2) = 0,0,1,2
    assert np.allclose(R[1, :3], [-0.09375 + 0.59375, 0.59375, -0.09375])
    # output 127: taps 62, 63, clamp(64)=63, clamp(65)=63
    assert np.allclose(R[127, 61:], [0.0, -0.09375, 0.59375 + 0.59375 - 0.09375])
    x = RNG.normal(size=(2, 6, 6, 3))
    assert np.allclose(O.resize_bicubic(t(x), 12).numpy(), S.resize_bicubic_legacy(x, 12), atol=1e-6)
    # 256 -> 128 (config 5): delta = 0 -> plain 2x subsampling
    R2 = O.bicubic_matrix(8, 4)
    assert np.array_equal(R2, np.eye(8)[::2])
    assert O.resize_bicubic(t(x), 6) is not None and O.resize_bicubic(t(x), 6).shape == (2, 6, 6, 3)


def test_avg_pool_same_counts_valid_elements():
    x = RNG.normal(size=(2, 2, 2, 3))
    assert np.allclose(O.avg_pool_same(t(x), 8).numpy(), x.mean(axis=(1, 2), keepdims=True))
    x = RNG.normal(size=(1, 5, 7, 2))
    assert np.allclose(O.avg_pool_same(t(x), 3).numpy(), S.avg_pool_same(x, 3))


def test_instance_norm_epsilon_outside_sqrt():
    x = t(RNG.normal(size=(2, 4, 4, 3)))
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    assert torch.allclose(O.instance_norm(x), (x - mu) / (var.sqrt() + 1e-5))
    assert not torch.allclose(O.instance_norm(x * 1e-4), (x * 1e-4 - mu * 1e-4) / torch.sqrt(var * 1e-8 + 1e-5))


def test_activation_tie_conventions():
    z = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    g, = torch.autograd.grad(torch.relu(z).sum(), z)
    assert g.tolist() == [0, 0, 0]                                   # tf.nn.relu slope 0 at 0
    x = t([-2.0, 3.0]).requires_grad_(True)
    g, = torch.autograd.grad(O.activation(x, 'lrelu').sum(), x)
    assert g.tolist() == [0.2, 1.0]
    lk = t(0.2).requires_grad_(True)
    y = O.prelu(x, lk)
    gl, = torch.autograd.grad(y.sum(), lk)
    assert y.tolist() == [-0.4, 3.0] and float(gl) == -2.0


def test_spectral_norm_matches_definition():
    W, u = RNG.normal(size=(3, 3, 4, 6)), RNG.normal(size=(1, 6))
    Wr = W.reshape(-1, 6)
    v = u @ Wr.T
    v = v / (np.sqrt((v ** 2).sum()) + 1e-12)
    u1 = v @ Wr
    u1 = u1 / (np.sqrt((u1 ** 2).sum()) + 1e-12)
    sigma = (v @ Wr @ u1.T)[0, 0]
    assert np.allclose(O.spectral_normed_weight(t(W), t(u)).numpy(), W / sigma)
    # one power iteration from a fixed u under-estimates the top singular value
    assert sigma <= np.linalg.svd(Wr, compute_uv=False)[0] + 1e-12


def test_rmsprop_matches_tf_kernel_definition():
    w, g = RNG.normal(size=20), RNG.normal(size=20)
    wt, mst, momt = t(w.copy()), torch.ones(20, dtype=torch.float64), torch.zeros(20, dtype=torch.float64)
    O.rmsprop_apply(wt, t(g), mst, momt, 2e-4)
    w2, ms2, mom2 = S.rmsprop(w, g, np.ones(20), np.zeros(20), 2e-4)
    assert np.allclose(wt.numpy(), w2) and np.allclose(mst.numpy(), ms2) and np.allclose(momt.numpy(), mom2)


def test_gradient_penalty_differentiates_the_tuple():
    """Quirk N10: tf.gradients((sigmoid(D), D), x) sums both outputs' gradients (edgegan.py:41 passes the
    discriminator's tuple; functional.py:27)."""
    cfg = O.Config(batch_size=2, df_dim=4, height=16, width=32)
    rng = np.random.RandomState(1)
    p = {}
    for s in O.discriminator_specs('D', cfg, 16, 32):
        p[s.name] = t(rng.normal(size=s.shape) * 0.2)
    real, fake = t(rng.normal(size=(2, 16, 32, 3))), t(rng.normal(size=(2, 16, 32, 3)))
    alpha = t([0.3, 0.8])
    got = O.gradient_penalty_term(p, 'D', fake, real, alpha, 10.0)
    x = (real + alpha.view(-1, 1, 1, 1) * (fake - real)).requires_grad_(True)
    sig, d = O.discriminator(p, 'D', x)
    g_both, = torch.autograd.grad(sig.sum() + d.sum(), x, retain_graph=True)
    g_logit, = torch.autograd.grad(d.sum(), x)
    want = 10.0 * torch.mean((g_both.flatten(1).norm(dim=1) - 1) ** 2)
    other = 10.0 * torch.mean((g_logit.flatten(1).norm(dim=1) - 1) ** 2)
    assert torch.allclose(got, want) and not torch.allclose(got, other)


def test_lazy_graph_matches_full_graph_and_substep_dependencies():
    """Like sess.run, a sub-step only evaluates what its loss depends on; the values must not change."""
    cfg = O.Config(batch_size=2, num_classes=0)
    rng = np.random.RandomState(2)
    vals = {}
    for s in O.variable_specs(cfg):
        kind, _, par = s.init.partition(':')
        vals[s.name] = (np.full(s.shape, float(par)) if kind == 'const' else rng.normal(size=s.shape) * 0.02)
    m = O.OracleModel(cfg, vals, dtype=torch.float32)
    inp = O.StepInputs(torch.tensor(rng.uniform(-1, 1, size=(2, 64, 128, 3)), dtype=torch.float32),
                       torch.tensor(rng.normal(size=(2, 100)), dtype=torch.float32),
                       torch.tensor(rng.uniform(size=(3, 2)), dtype=torch.float32), torch.tensor([0.3]))
    full = O.forward_graph(m.p, inp, cfg, 0.3)
    part = O.forward_graph(m.p, inp, cfg, 0.3, want=['zl_loss'])
    assert set(part) == {'zl_loss', 'z_recon'} and torch.allclose(part['zl_loss'], full['zl_loss'])
    assert m.substep_names() == [n for n in O.SUBSTEPS if n != 'd_optim2']      # edgegan.py:115-116


def test_bf16_operand_emulation_rounds_each_bilinear_map():
    """oracle.bf16_operands(): forward, input gradient and filter gradient of a qualifying contraction each use
    bf16-rounded operands (round-to-nearest-even) with exact products and sums; non-qualifying shapes are untouched;
    the backward maps stay differentiable (needed for the gradient-penalty double backward)."""
    import torch
    torch.manual_seed(0)
    r = lambda t: t.detach().float().bfloat16().double()
    x = torch.randn(2, 6, 6, 64, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 3, 64, 32, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(2, 6, 6, 32, dtype=torch.float64)
    with O.bf16_operands():
        y = O.conv2d(x, w, 1, 'SAME')
        gx, gw = torch.autograd.grad((y * dy).sum(), [x, w])
    assert torch.equal(y, O.conv2d(r(x), r(w), 1, 'SAME'))
    assert torch.equal(gx, torch.autograd.grad((O.conv2d(x, r(w), 1, 'SAME') * r(dy)).sum(), x)[0])
    assert torch.equal(gw, torch.autograd.grad((O.conv2d(r(x), w, 1, 'SAME') * r(dy)).sum(), w)[0])
    # 0.1 is not a bf16 number: round-to-nearest-even gives 0.10009765625
    assert float(r(torch.tensor([0.1], dtype=torch.float64))[0]) == 0.10009765625
    # a 3-channel input does not qualify (contracted channels % 32 != 0): bit-identical to the exact path
    x3 = torch.randn(1, 5, 5, 3, dtype=torch.float64)
    w3 = torch.randn(4, 4, 3, 64, dtype=torch.float64)
    with O.bf16_operands():
        y3 = O.conv2d(x3, w3, 2, 'SAME')
    assert torch.equal(y3, O.conv2d(x3, w3, 2, 'SAME'))
    # transposed conv: forward qualifies on (Cin, Cout), its filter gradient on (Cout, Cin) as the engine launches it
    xd = torch.randn(1, 4, 4, 64, dtype=torch.float64, requires_grad=True)
    wd = torch.randn(5, 5, 32, 64, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(32, dtype=torch.float64)
    with O.bf16_operands():
        yd = O.deconv2d(xd, wd, b, (8, 8))
    assert torch.equal(yd, O.deconv2d(r(xd), r(wd), b, (8, 8)))
    # double backward through the emulated maps
    with O.bf16_operands():
        g1 = torch.autograd.grad(O.conv2d(x, w, 1, 'SAME').tanh().sum(), x, create_graph=True)[0]
        g2 = torch.autograd.grad((g1 ** 2).sum(), w)[0]
    assert torch.isfinite(g2).all() and float(g2.abs().max()) > 0
    assert not O._BF16_OPERANDS
