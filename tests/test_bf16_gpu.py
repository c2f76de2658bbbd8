"""compute_dtype='bf16' (BASELINE configs[1], [3], [4]): dense contractions on the bf16 MFMA with fp32 accumulation
and fp32 storage.  The oracle emulates the engine's operand rounding (oracle/edgegan_oracle.py: bf16_operands), so

  * per-op results (forward, input gradient, filter gradient of every conv / transposed-conv case of the fp32 op
    tests) must agree to the fp32-ACCUMULATION tolerances (2e-5 / 5e-5 max-norm): the products of bf16 numbers are
    exact in fp32, only the summation order differs;
  * network forwards agree to 2e-2 max-norm (measured <= 6e-3, the worst being the discriminators' scalar logit, a
    16384-term sum with cancellation): fp32-vs-fp64 noise moves ~1e-4 of the rounded activations across a bf16
    rounding boundary (a 2^-8 relative change of that element);
  * sub-step gradients: relative L2 <= max(0.1, 3 x the oracle's own fp32-vs-fp64 self-consistency under the same
    emulation).  Measured on MI355X (batch 2, 14 classes): D sub-steps 0.03-0.05, classifier median 0.013, generator
    sub-steps 0.03-0.06, encoder sub-step 0.25 -- against 0.32 for the oracle's fp32 run of itself, i.e. the bf16
    noise floor of this graph at random init, not a kernel property (the same kernels pass the op tests at 2e-6);
  * losses agree to 1e-2 (the gradient-penalty term is a norm of a bf16-computed input gradient).
"""
import os

import numpy as np
import pytest
import torch

import edgegan_amd as ea
import edgegan_oracle as O
import parity

pytestmark = pytest.mark.gpu


def _assert(rows):
    bad = [r for r in rows if not r[1] <= r[2]]
    assert not bad, '\n' + parity.format_rows(bad)


def test_bf16_ops_match_rounded_operand_oracle():
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='bf16'))
    rows = parity.bf16_op_checks(net)
    # The discriminators' 4x4 stride-2 forward layers in the space-to-depth halo form, ONE-plane instantiation
    # (k_gemm_x3h.hip, NPROD = 1, S2D): un-split, and split K with several K chunks per block -- the combination that gave
    # deterministic wrong results in round 3 (a dead look-ahead fragment read landing on a re-used register, DESIGN.md
    # section 3.10), at the shapes of tools/gpu_s2d_check.py plus 8 images of 4x8 outputs per tile.
    for case in [('Dp conv_1 4x4s2 64->128 @64 nb64 (s2d halo)', 64, 64, 64, 64, 128, 4, 2, 0, False),
                 ('Dp conv_3 4x4s2 128->256 @32 nb64 (s2d halo, split K, 8 chunks)', 64, 32, 32, 128, 256, 4, 2, 0, False),
                 ('Dp conv_3 4x4s2 128->256 @32 nb4 (s2d halo, split K, 2 chunks)', 4, 32, 32, 128, 256, 4, 2, 0, False),
                 ('Dp conv_4 4x4s2 256->512 @16 nb64 (s2d halo, split K)', 64, 16, 16, 256, 512, 4, 2, 0, False),
                 ('D conv_4 4x4s2 256->512 8x16 nb64 (s2d halo, 8 images per tile)', 64, 8, 16, 256, 512, 4, 2, 0, False)]:
        rows += parity.check_conv(net, case, seed=7, bf16=True)
    assert len(rows) >= 135
    _assert(rows)
    net.close()


def test_bf16_fused_mru_chains():
    """The fused forms of the MRU unit on the plane-fed kernels (Conv_1's PReLU and its bf16 plane from conv_bf16p's epilogue,
    K-split slabs summed by their consumer) against the oracle with the engine's operand rounding emulated; a chain's
    tolerance allows a rounding-boundary flip of an intermediate value (2^-8 of that element)."""
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='bf16'))
    _assert(parity.fused_chain_checks(net, bf16=True))
    net.close()


@pytest.mark.parametrize('num_classes', [14, 0])
def test_bf16_network_forwards(num_classes):
    cfg = ea.Config(batch_size=2, num_classes=num_classes, compute_dtype='bf16')
    values = ea.init_variables(cfg, 1)
    net = ea.EdgeGAN(cfg, values)
    _assert(parity.net_checks(net, values, bf16=True, tol=2e-2))
    net.close()


def test_bf16_substeps_14_classes():
    """configs[3]'s graph and arithmetic (14-class 64x128, bf16 contractions) at batch 2."""
    cfg = ea.Config(batch_size=2, compute_dtype='bf16')
    values = ea.init_variables(cfg, 0)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg), grad_tol=0.1, bf16=True, loss_tol=1e-2)
    assert len(rows) > 400
    # A scalar PReLU slope's gradient is a cancelling sum over every element of its layer; where it is < 1 % of the
    # sub-step's largest gradient, the bf16 noise of the other 99 % lands on it in ABSOLUTE terms (measured: the unit-2
    # `Conv_1` slope, |g| = 9e-3 gmax, deviates by 0.12 - 0.16 of itself depending on the contraction kernels' summation
    # order, i.e. by 1.4e-3 gmax, while full-size variables deviate by up to 0.05 gmax).  Such scalars are judged by
    # their absolute error against 5e-3 of the sub-step's largest gradient (the same rule as tests/test_x3_gpu.py).
    judged = []
    for name, err, tol in rows:
        if err > tol and 'prelu/param' in name and '|g|/gmax' in name:
            frac = float(name.split('|g|/gmax')[1].strip(' ]'))
            judged.append((name + ' (absolute: err * |g|/gmax vs 5e-3)', err * max(frac, 1e-5), 5e-3))
        else:
            judged.append((name, err, tol))
    _assert(judged)
    net.close()


@pytest.mark.parametrize('name,B,H,W', [('d_optim', 16, 64, 128), ('d_optim2', 16, 64, 128), ('g_optim_u', 16, 64, 128),
                                        ('d_optim2', 2, 256, 512), ('d_optim_patch3', 16, 64, 128), ('d_optim2', 64, 64, 128),
                                        ('d_optim', 2, 256, 512), ('g_optim_u', 2, 256, 512)],
                         ids=['d_optim-b16', 'd_optim2-b16', 'g_optim_u-b16', 'd_optim2-256x512-b2', 'd_optim_patch3-b16', 'd_optim2-b64',
                              'd_optim-256x512-b2', 'g_optim_u-256x512-b2'])
def test_bf16_substep_at_batch16_against_the_emulating_oracle(name, B, H, W):
    """the bf16 mode against the EMULATING oracle (operands of every qualifying contraction rounded to
    bf16 exactly as the engine rounds them, everything else fp64: edgegan/models/edgegan.py:109-116 evaluated under
    O.bf16_operands()) at 64x128, 14 classes, batch 16 -- the joint discriminator's sub-step (gradient penalty: forward-mode
    tangent + one reverse sweep through the plane-fed kernels, phases, split K) and the classifier's (54 % of the iteration's
    FLOPs, the producer-written planes of PReLU / gate / pool / min-max).  At this batch the plane-fed kernels run their
    production tile forms, which the batch-2 test above never reaches.  Every optimised variable: relative L2 <=
    max(3e-2, 3 x the oracle's own fp32-vs-fp64 deviation under the same emulation) -- the bf16 noise floor of this graph at
    random init (an fp32 evaluation of the ORACLE'S OWN emulation deviates from its fp64 evaluation by about as much: values
    that get rounded sit on the other side of a bf16 rounding boundary in ~1e-4 of the cases).  Measured on MI355X
    (profiles/r04m_pytest.txt): d_optim median 3.2e-2, max 3.8e-2 over its 6 variables; d_optim2 median 3.4e-2 over 57
    variables, scalar PReLU slopes judged in absolute terms (below).  A wiring / stale-plane bug shows as >= 3e-1.
    Round 5: the generators' sub-step g_optim_u at batch 16 (all four critics' input-gradient chains, the
    transposed convolutions' phases) and the classifier's sub-step at configs[4]'s geometry (256x512, batch 2: 128x128 ...
    16x16 maps, 2^2 more pixels per image than any 64x128 test) against the same emulating oracle; later in round 5 the edge
    discriminator's sub-step (bicubic resize in front of the stacked passes) at batch 16 and the classifier's sub-step at configs[3]'s
    full per-GPU batch of 64.  Round 6: the joint discriminator's and the generators' sub-steps at configs[4]'s
    geometry (256x512, batch 2) -- an oracle comparison where the full-size test below only has the f32x3 engine to compare with.
    (The encoder's sub-step is not in this list on purpose: at random init the emulating oracle's OWN
    fp32-vs-fp64 deviation on it is 0.2 ... 0.3 for most filters -- bf16 rounding moves the signs inside its L1 loss and the ReLU
    decisions of its 4 ... 16-pixel InstanceNorm layers -- so a tolerance derived from it proves nothing; measured engine error 0.21
    median.  Round 6 looked for a better-conditioned state: after 10 / 25 / 50 / 100 training iterations of the fp32 oracle the
    emulated e_optim's self-deviation stays at a median 0.07 ... 0.08 and a maximum 0.3 while the un-emulated fp32 one falls to 1e-4
    (tools/cpu_eoptim_conditioning.py, profiles/r06_eoptim_conditioning.txt): the sensitivity is to the bf16 rounding boundaries of
    the operands, which training does not move away from.  Its bf16 coverage is the batch-2 test above and the property tests below.)"""
    import full_batch_lib as FB
    import test_tight_gpu as TT
    key = (f'bf16_b{B}/' if (H, W) == (64, 128) else f'bf16_{H}x{W}_b{B}/') + name
    case = FB.CASE[key]
    cfg = ea.Config(compute_dtype='bf16', **case[2])
    values = ea.init_variables(cfg, case[3])
    b = ea.synthetic_batch(cfg, seed=case[4])
    live = bool(os.environ.get('EG_LIVE_ORACLE'))
    if live:      # the emulating oracle evaluated here and now (minutes of CPU time): exact relative L2 on every variable
        with O.bf16_operands():
            om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
            inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
            g64, l64 = om.compute_grads(name, inp, inp.eps[0])
            om32 = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float32)
            inp32 = O.StepInputs(*[torch.tensor(np.asarray(b[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
            g32, _ = om32.compute_grads(name, inp32, inp32.eps[0])
        gmax = max(float(g.abs().max()) for g in g64.values())
        var_tol = {vn: 3.0 * parity.rel_l2(g32[vn].double().numpy(), g.numpy(), 1e-5 * gmax) for vn, g in g64.items()}
        scalars = {vn: float(g) for vn, g in g64.items() if g.numel() == 1}
    else:         # the committed digests of exactly that evaluation (tests/full_batch_lib.py, tests/golden/make_full_batch_goldens.py)
        fb = FB.load(key)
        gmax = fb['gmax']
        var_tol = {vn: 3.0 * d for vn, d in fb['self_dev'].items()}
        scalars = {vn: float(e['exact'][0]) for vn, e in fb['vars'].items() if 'exact' in e and e['exact'].size == 1}
    floor = 3e-2 if (H, W) == (64, 128) else 4e-2     # (256x512, batch 2: measured 3.3e-2 on one gate bias -- 512 pixels per column sum)
    net = ea.EdgeGAN(cfg, values)
    net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
    if live:
        rows = TT._grad_rows(net, None, None, name, floor, f'bf16 B={B} {H}x{W}', var_tol=var_tol, precomputed=(g64, l64))
    else:
        rows = TT._grad_rows_digest(net, name, fb, floor, f'bf16 B={B} {H}x{W}', var_tol=var_tol)
    # a scalar PReLU slope's gradient is ONE cancelling sum over its whole layer: where it is a small fraction of the
    # sub-step's largest gradient the bf16 noise of everything else lands on it in ABSOLUTE terms -- judged, as in
    # test_bf16_substeps_14_classes, by its absolute error against 5e-3 of the sub-step's largest gradient
    judged = []
    for n, e, t in rows:
        vn = n.split(' grad ')[-1].split(' [')[0] if ' grad ' in n else None
        if vn in scalars and e > t:
            h = float(net.get_variable(vn + '/grad').reshape(-1)[0])
            judged.append((n + ' (scalar: |err| / gmax vs 5e-3)', abs(h - scalars[vn]) / gmax, 5e-3))
        else:
            judged.append((n, e, 1e-2 if ' loss ' in n else t))      # losses: the bf16 tolerance of this file
    rows = judged
    net.close()
    if name == 'd_optim2':
        assert len(rows) >= 55
    grad_errs = sorted(e for n, e, t in rows if ' grad ' in n)
    print(name, f'bf16 B={B} {H}x{W} vs emulating oracle: median %.1e max %.1e over %d variables' % (grad_errs[len(grad_errs) // 2], grad_errs[-1], len(grad_errs)))
    _assert(rows)


def test_bf16_inference_batch64_single_class():
    """configs[1]: single-class 64x128, batch 64, generator-only inference (edgegan.test path) in bf16 -- against the
    emulating oracle (tight) and against the plain fp64 oracle (the bf16 deviation a user sees)."""
    cfg = ea.Config(batch_size=64, num_classes=0, compute_dtype='bf16')
    values = ea.init_variables(cfg, 3)
    net = ea.EdgeGAN(cfg, values)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    imgs = ea.synthetic_batch(cfg, seed=11)['images']
    e, i = net.infer(imgs, None, eps2=[0.3, -0.7])
    with O.bf16_operands():
        e_emu, i_emu = om.infer(parity.t64(imgs), None, 0.3, -0.7)
    e_ref, i_ref = om.infer(parity.t64(imgs), None, 0.3, -0.7)
    for got, emu, ref in ((e, e_emu, e_ref), (i, i_emu, i_ref)):
        assert parity.rel_l2(got, emu.numpy()) < 5e-3, parity.rel_l2(got, emu.numpy())     # same rounding, fp32 sums
        assert parity.relerr(got, emu.numpy()) < 3e-2, parity.relerr(got, emu.numpy())     # worst of 786k pixels
        assert parity.rel_l2(got, ref.numpy()) < 3e-2, parity.rel_l2(got, ref.numpy())     # bf16 vs exact arithmetic
    net.close()


@pytest.mark.parametrize('flags', [dict(), dict(E_norm='batch', G_norm='batch'), dict(E_norm=None, G_norm=None)],
                         ids=['default', 'batch_norms', 'no_norms'])
def test_bf16_inference_norm_flags(flags):
    """The test graph under E_norm / G_norm = 'batch' and None (normalization.py:19-27) in bf16 mode: the forward-only graph's plane
    hand-overs (a block norm that is not InstanceNorm leaves the plane to its consumer) against the emulating oracle."""
    cfg = ea.Config(batch_size=8, compute_dtype='bf16', **flags)
    values = ea.init_variables(cfg, 4)
    net = ea.EdgeGAN(cfg, values)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    imgs = ea.synthetic_batch(cfg, seed=12)['images']
    cls = (np.arange(8) % 14).astype(np.int32)
    e, i = net.infer(imgs, cls, eps2=[0.3, -0.7])
    with O.bf16_operands():
        e_emu, i_emu = om.infer(parity.t64(imgs), torch.tensor(cls), 0.3, -0.7)
    e_ref, i_ref = om.infer(parity.t64(imgs), torch.tensor(cls), 0.3, -0.7)
    errs = [parity.rel_l2(got, emu.numpy()) for got, emu in ((e, e_emu), (i, i_emu))]
    refs = [parity.rel_l2(got, ref.numpy()) for got, ref in ((e, e_ref), (i, i_ref))]
    print('bf16 inference', flags, 'vs emulating oracle %.2e %.2e; vs the fp64 oracle %.2e %.2e' % (*errs, *refs))
    # (same operand rounding, other fp32 summation order: a sum that lands on the other side of a bf16 rounding boundary moves the
    #  next layer's operand by one bf16 ulp, and an 8-row BatchNorm spreads that over the batch -- measured 2e-3 ... 6e-3; a wrong
    #  hand-over shows as O(1))
    assert max(errs) < 1e-2, errs
    assert max(refs) < 3e-2, refs
    net.close()


def test_bf16_train_step_is_deterministic_and_finite():
    cfg = ea.Config(batch_size=8, compute_dtype='bf16')
    values = ea.init_variables(cfg, 0)
    b = ea.synthetic_batch(cfg)
    outs = []
    for _ in range(2):
        net = ea.EdgeGAN(cfg, values)
        net.upload_batch(**b)
        for _ in range(3):
            net.train_step()
        outs.append((net.losses(), net.get_variable('E/e_resnet_128_1/res1/conv2d/w'), net.get_variable('D/d_conv_3/conv2d/w')))
        net.close()
    assert all(np.isfinite(v) for v in outs[0][0].values())
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize('shape', [(64, 64, 128), (16, 256, 512)], ids=['configs3_b64_64x128', 'configs4_b16_256x512'])
def test_bf16_full_size_gradients_track_fp32(shape):
    """BASELINE's full sizes AS WRITTEN (configs[3]: batch 64 per GPU at 64x128; configs[4]: batch 16 per GPU at 256x512),
    14 classes, bf16: every sub-step's gradients from the bf16 engine against the fp32 engine (f32x3 arithmetic: fp32
    results, tests/test_x3_gpu.py) on the same weights and batch; the bf16 sub-step is run twice and must reproduce itself
    bit for bit; a whole bf16 training iteration on top must leave finite losses.  The batch-2 tests above never reach the
    production shapes (256-pixel tiles, tens of K steps per weight-gradient block, 84-way split reductions, 2^31-byte
    activation extents at 256x512); a kernel bug there shows up as O(1) error, bf16 rounding as a few percent (the encoder
    sub-step ~0.3, see the module docstring)."""
    B, H, W = shape
    cfg32 = ea.Config(batch_size=B, height=H, width=W, compute_dtype='f32x3')
    cfg16 = ea.Config(batch_size=B, height=H, width=W, compute_dtype='bf16')
    values = ea.init_variables(cfg32, 0)
    batch = ea.synthetic_batch(cfg32)
    n32, n16 = ea.EdgeGAN(cfg32, values), ea.EdgeGAN(cfg16, values)
    names = {1: ['D/d_conv_1/conv2d/w', 'D/d_conv_4/conv2d/w'], 2: ['D_patch2/d_conv_3/conv2d/w', 'D_patch2/d_conv_0/conv2d/w'],
             3: ['D_patch3/d_conv_4/conv2d/w'],
             4: ['D2/mru_conv_unit_t_1_layer_0/Conv_2/weights', 'D2/mru_conv_unit_t_3_layer_0/Conv_1/weights',
                 'D2/mru_conv_unit_t_4_layer_0/update_gate/weights'],
             5: ['G1/g_dconv_1/deconv2d/w', 'G2/g_dconv_3/deconv2d/w', 'G2/g_lin_0/Matrix'],
             6: ['E/e_resnet_128_1/res1/conv2d/w', 'E/e_resnet_512_4/res2/conv2d/w', 'E/FC8_mu/w'],
             7: ['G1/g_dconv_2/deconv2d/w', 'G2/g_dconv_1/deconv2d/w']}
    worst = {}
    for net in (n32, n16):
        net.upload_batch(**batch)
    for k in range(1, 8):
        n32.substep(k, apply=False)
        n16.substep(k, apply=False)
        first = {vn: n16.get_variable(vn + '/grad') for vn in names[k]}
        n16.substep(k, apply=False)
        for vn in names[k]:
            a, b = n16.get_variable(vn + '/grad'), n32.get_variable(vn + '/grad')
            assert np.all(np.isfinite(a))
            assert np.array_equal(a, first[vn]), f'sub-step {k} {vn}: the bf16 engine does not reproduce itself'
            worst[(k, vn)] = parity.rel_l2(a, b)
    n32.close()
    n16.train_step()
    losses = n16.losses()
    n16.close()
    assert all(np.isfinite(v) for v in losses.values()), losses
    bad = {kv: e for kv, e in worst.items() if e > (0.7 if kv[0] == 6 else 0.25)}
    assert not bad, (bad, worst)
