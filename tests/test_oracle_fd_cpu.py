"""An independent derivation of the oracle's gradients: fp64 central finite differences.

Every gradient the parity tests compare the HIP path against comes from torch autograd over oracle/edgegan_oracle.py
(incl. the create_graph double backward of the WGAN-GP term).  Here the same gradients are derived WITHOUT autograd's
backward rules: for a random direction v over the optimised variables,
        (L(theta + h v) - L(theta - h v)) / 2h   must equal   <grad_autograd L, v>,
for the four kinds of sub-step loss (edgegan/models/edgegan.py:277-342):
  * joint_dis_dloss  = WGAN + gradient penalty through the discriminator's TUPLE (quirk N10, edgegan.py:38-42,
    functional.py:26-29) -- checks the double backward; the penalty's INNER gradient d/dx [sigmoid(D)+D] is checked
    against finite differences over the input as well, so no link of the chain rests on autograd alone;
  * loss_d_ac        = focal AC loss through spectral norm, PReLU and the min-max gate (conv.py:190-198,
    normalization.py:46-70, functional.py:5-16);
  * zl_loss          = L1 latent reconstruction through G1 and the encoder (edgegan.py:337-342);
  * edge_gloss / image_gloss w.r.t. the generators, through all four critics (edgegan.py:311-323).
gf = df = 8 keeps it to seconds (the classifier and the encoder have fixed widths).  Tolerance 1e-6 relative.
Step size: the losses are only piecewise smooth, and the gradient penalty is even DISCONTINUOUS in theta (the
penalised input-gradient changes by a finite amount when a LeakyReLU pre-activation changes sign), so a central
difference is only meaningful when no kink lies inside [theta - hv, theta + hv]: measured on the joint discriminator,
h = 1e-4 is garbage, 1e-5 agrees to 7e-5, 1e-6 to 2e-10, 1e-7 to 9e-9 (rounding).  Each direction is therefore
evaluated at h = 1e-6, 1e-7 and 1e-8 (relative to each variable's scale) and the best agreement is asserted, with the
difference quotient's own rounding noise (200 eps |L| / h) allowed on top of the relative tolerance."""
import numpy as np
import pytest
import torch

import edgegan_amd as ea
import edgegan_oracle as O
import parity

DT = torch.float64
TOL = 1e-6


@pytest.fixture(scope='module')
def setup():
    cfg = ea.Config(batch_size=2, gf_dim=8, df_dim=8)
    values = ea.init_variables(cfg, 3)
    rng = np.random.RandomState(5)
    for k, v in values.items():          # move every zero-initialised bias off zero so that its direction matters
        if v.size and not np.any(v):
            values[k] = (rng.normal(size=v.shape) * 0.02).astype(np.float32)
    b = ea.synthetic_batch(cfg, seed=17)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=DT)
    inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
    return cfg, om, inp


def _loss(om, inp, key, p):
    return O.forward_graph(p, inp, om.cfg, inp.eps[0], want=[key])[key]


def _directional(om, inp, loss_key, names, seed, h_rels=(1e-6, 1e-7, 1e-8)):
    """([finite-difference directional derivatives, one per step size], <autograd grad, v>) along one random direction
    over `names`."""
    p = om._leaf_params()
    loss = _loss(om, inp, loss_key, p)
    grads = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    g = torch.Generator().manual_seed(seed)
    ana, vs = 0.0, {}
    for n, gr in zip(names, grads):
        w = om.p[n]
        scale = float(w.abs().max()) if float(w.abs().max()) > 0 else 0.02
        vs[n] = torch.randn(w.shape, generator=g, dtype=DT) * scale
        if gr is not None:
            ana += float((gr * vs[n]).sum())
    fds = []
    l0 = abs(float(loss.detach()))
    for h in h_rels:
        pp, pm = dict(om.p), dict(om.p)
        for n, v in vs.items():
            pp[n] = om.p[n] + h * v
            pm[n] = om.p[n] - h * v
        with torch.enable_grad():        # the penalty's inner gradient needs the graph even for a plain evaluation
            fd = (float(_loss(om, inp, loss_key, pp).detach()) - float(_loss(om, inp, loss_key, pm).detach())) / (2 * h)
            fds.append((fd, 200 * 2.2e-16 * l0 / h))
    return fds, ana


def _check(om, inp, loss_key, groups):
    bad = []
    for seed, names in enumerate(groups):
        fds, ana = _directional(om, inp, loss_key, names, seed)
        err = min(max(abs(fd - ana) - noise, 0.0) / max(abs(fd), abs(ana), 1e-30) for fd, noise in fds)
        if not err <= TOL:
            bad.append((loss_key, names[:2], fds, ana, err))
    assert not bad, bad


def _trainable(om, prefix):
    return [n for n in om.p if n.startswith(prefix + '/') and om.trainable[n]]


def test_gradient_penalty_double_backward_matches_finite_differences(setup):
    cfg, om, inp = setup
    for loss_key, prefix in (('joint_dis_dloss', 'D'), ('image_dis_dloss', 'D_patch2')):
        allv = _trainable(om, prefix)
        _check(om, inp, loss_key, [allv, allv, [f'{prefix}/d_conv_0/conv2d/w'], [f'{prefix}/d_conv_3/conv2d/w'],
                                   [f'{prefix}/d_conv_4/conv2d/w'], [f'{prefix}/d_linear_5/Matrix'], [f'{prefix}/d_linear_5/bias']])


def test_penalty_inner_gradient_matches_finite_differences_over_the_input(setup):
    """g = d/dx [sum_b sigmoid(D(x_b)) + D(x_b)] (the tuple, quirk N10) along random image directions."""
    cfg, om, inp = setup
    x = inp.images.clone().requires_grad_(True)
    sig, logit = O.discriminator(om.p, 'D', x)
    f = lambda t: float(sum(o.sum() for o in O.discriminator(om.p, 'D', t)))
    g, = torch.autograd.grad(sig.sum() + logit.sum(), x)
    gen = torch.Generator().manual_seed(0)
    for _ in range(3):
        v = torch.randn(x.shape, generator=gen, dtype=DT)
        ana = float((g * v).sum())
        errs = []
        for h in (1e-6, 1e-7, 1e-8):
            fd = (f(inp.images + h * v) - f(inp.images - h * v)) / (2 * h)
            errs.append(abs(fd - ana) / max(abs(fd), abs(ana)))
        assert min(errs) <= TOL, (errs, ana)
    # and the logit alone would give a different vector: the sigmoid term is really in there
    g_logit, = torch.autograd.grad(O.discriminator(om.p, 'D', x)[1].sum(), x)
    assert float((g - g_logit).abs().max()) > 1e-3 * float(g.abs().max())


def test_classifier_loss_through_spectral_norm_and_minmax_gate_matches_finite_differences(setup):
    cfg, om, inp = setup
    allv = [n for n in _trainable(om, 'D2') if not n.startswith('D2/Conv_1/')]       # dead head excluded (N8)
    u = 'D2/mru_conv_unit_t_%d_layer_0'
    _check(om, inp, 'loss_d_ac', [allv, allv, [u % 2 + '/update_gate/weights'], [u % 1 + '/update_gate/biases'],
                                  [u % 3 + '/Conv_1/prelu/param'], [u % 1 + '/norm_activation_in/prelu/param'],
                                  [u % 4 + '/Conv_2/weights'], [u % 2 + '/Conv_3/weights'], ['D2/Conv/weights'],
                                  ['D2/fully_connected/weights'], ['D2/fully_connected/biases']])


def test_latent_l1_loss_through_generator_and_encoder_matches_finite_differences(setup):
    cfg, om, inp = setup
    allv = _trainable(om, 'E')
    _check(om, inp, 'zl_loss', [allv, allv, ['E/e_resnet_64_0/conv2d/w'], ['E/e_resnet_512_4/res2/conv2d/w'],
                                ['E/e_resnet_128_1/shortcut/conv2d/b'], ['E/FC8_sigma/w'], ['E/FC8_mu/b']])


def test_generator_losses_through_all_critics_match_finite_differences(setup):
    cfg, om, inp = setup
    g1, g2 = _trainable(om, 'G1'), _trainable(om, 'G2')
    _check(om, inp, 'edge_gloss', [g1, ['G1/g_dconv_4/deconv2d/w'], ['G1/batch_norm/BatchNorm/gamma'], ['G1/g_lin_0/Matrix']])
    _check(om, inp, 'image_gloss', [g2, ['G2/g_dconv_1/deconv2d/w'], ['G2/batch_norm/BatchNorm/beta'], ['G2/g_dconv_4/deconv2d/b']])
