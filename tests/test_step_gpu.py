"""GPU parity of the networks, the 7 optimizer sub-steps (gradients incl. the WGAN-GP double
backward, losses, RMSProp application, BatchNorm moving statistics) and inference."""
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


@pytest.mark.parametrize('num_classes', [14, 0])
def test_network_forwards(num_classes):
    cfg = ea.Config(batch_size=2, num_classes=num_classes)
    values = ea.init_variables(cfg, 1)
    net = ea.EdgeGAN(cfg, values)
    _assert(parity.net_checks(net, values))
    net.close()


def test_substeps_14_classes():
    """config[2]'s graph (14-class 64x128 fp32 training) at batch 2."""
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg), share_oracle=True)   # (re-run in f32x3: test_x3_gpu.py)
    assert len(rows) > 400
    _assert(rows)
    net.close()


def test_substeps_odd_batch():
    """Batch 3: pixel counts that are not multiples of the 64 / 128-row tiles in every layer (ragged last tiles,
    partial split-K slabs, odd BatchNorm row counts)."""
    cfg = ea.Config(batch_size=3)
    values = ea.init_variables(cfg, 11)
    net = ea.EdgeGAN(cfg, values)
    _assert(parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=3), share_oracle=True))
    net.close()


def test_substeps_two_iterations():
    """SURVEY 8c golden item (iii): two consecutive iterations -- the second one starts from updated weights,
    non-initial RMSProp slots and advanced BatchNorm moving statistics.  (Fixture note: with variable seed 21 /
    batch seed 8 the encoder sub-step of iteration 0 sits on a ReLU kink -- the engine deviates from the fp64 oracle
    by 2.8e-2 in every encoder variable, and so does the ORACLE'S OWN fp32 evaluation, 2.9e-2 -- which is the
    conditioning effect DESIGN.md section 4 describes, not a kernel property; this fixture avoids it.)"""
    cfg = ea.Config(batch_size=2, num_classes=0)
    values = ea.init_variables(cfg, 22)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=9), iterations=2)
    assert any(r[0].startswith('it1 ') for r in rows)
    _assert(rows)
    net.close()


def test_substeps_single_class_batch4():
    """config[0]: single-class 64x128, batch 4 (edgegan.train --nomulticlasses)."""
    cfg = ea.Config(batch_size=4, num_classes=0)
    values = ea.init_variables(cfg, 5)
    net = ea.EdgeGAN(cfg, values)
    _assert(parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=99), share_oracle=True))
    net.close()


@pytest.mark.parametrize('flags', [dict(use_image_discriminator=False), dict(use_edge_discriminator=False),
                                   dict(use_image_discriminator=False, use_edge_discriminator=False, num_classes=0),
                                   dict(image_dis_size=64, edge_dis_size=96)],
                         ids=['no_image_dis', 'no_edge_dis', 'joint_only_single_class', 'unequal_dis_sizes'])
def test_substeps_non_default_discriminator_flags(flags):
    """use_image_discriminator / use_edge_discriminator = False (edgegan/train.py:61-66): the network is not built, its
    sub-step is not registered and its loss terms are the constant 0 (models/edgegan.py:111-114,164-174,253-274,288-311);
    image_dis_size != edge_dis_size (train.py:63-67).  Every remaining sub-step against the oracle, which applies the same
    rules; a switched-off network has no variables on either side."""
    cfg = ea.Config(batch_size=3, **flags)
    names = {s.name for s in ea.variable_specs(cfg)}
    assert any(n.startswith('D_patch2/') for n in names) == cfg.use_image_discriminator
    assert any(n.startswith('D_patch3/') for n in names) == cfg.use_edge_discriminator
    values = ea.init_variables(cfg, 7)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=31))
    ran = {r[0].split()[3] for r in rows if r[0].startswith('it0 substep')}
    assert ('d_optim_patch2' in ran) == cfg.use_image_discriminator and ('d_optim_patch3' in ran) == cfg.use_edge_discriminator
    _assert(rows)
    # a whole concurrent iteration (lanes on) with the switched-off sub-steps skipped: finite, and the skipped terms read 0
    net.set_variables(values)
    net.train_step()
    l = net.losses()
    assert all(np.isfinite(v) for v in l.values()), l
    if not cfg.use_image_discriminator:
        assert l['image_dis_dloss'] == 0.0 and l['image_dis_gloss'] == 0.0
    if not cfg.use_edge_discriminator:
        assert l['edge_dis_dloss'] == 0.0 and l['edge_dis_gloss'] == 0.0
    ev = net.evaluate(tensors=True)
    assert ('resized_inputs' in ev) == cfg.use_image_discriminator and ('resized_edges' in ev) == cfg.use_edge_discriminator
    net.close()


@pytest.mark.parametrize('flags', [dict(G_norm='batch'), dict(E_norm='batch'), dict(D_norm='batch'),
                                   dict(E_norm='batch', G_norm='batch', D_norm='batch', num_classes=0),
                                   dict(E_norm=None, G_norm=None, D_norm=None), dict(E_norm='batch', G_norm=None, D_norm=None)],
                         ids=['G_batch', 'E_batch', 'D_batch', 'all_batch_single_class', 'all_None', 'E_batch_G_D_None'])
def test_substeps_batch_norm_flags(flags):
    """E_norm / G_norm / D_norm = 'batch' (edgegan/train.py:54-59, nn/modules/normalization.py:19-25): contrib batch_norm with
    learned scale / shift in the encoder's residual blocks, the generators' transposed-convolution blocks and blocks 2 - 4 of
    the three discriminators -- there including its forward tangent and the reverse of that tangent, which the gradient
    penalty's second derivative runs through (functional.py:26-29) -- and None (normalization.py:26-27: the block keeps its
    activation only; reachable from Python, the string flag's 'None' fails line 12's assertion).  Every sub-step against the fp64 oracle (autograd's
    double backward through torch's own arithmetic of the same formula), gradients of beta / gamma included; then a whole
    concurrent iteration and the post-update evaluation."""
    cfg = ea.Config(batch_size=3, **flags)
    names = {s.name for s in ea.variable_specs(cfg)}
    assert ('G1/g_dconv_1/batch_norm/BatchNorm/gamma' in names) == (cfg.G_norm == 'batch')
    values = ea.init_variables(cfg, 11)
    rng = np.random.RandomState(5)
    for n in values:      # (beta = 0, gamma = 1 at initialisation: move them so that a wrong scale / shift gradient shows)
        if n.endswith('BatchNorm/gamma'):
            values[n] = (1.0 + 0.2 * rng.standard_normal(values[n].shape)).astype(np.float32)
        elif n.endswith('BatchNorm/beta'):
            values[n] = (0.1 * rng.standard_normal(values[n].shape)).astype(np.float32)
    net = ea.EdgeGAN(cfg, values)
    # (self_consistency: per-variable tolerance max(1e-2, 3 x the oracle's own fp32-vs-fp64 deviation) -- e_optim's L1 / ReLU kinks
    #  at a random initialisation, parity.substep_checks)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=37), self_consistency=True)
    checked = {r[0] for r in rows}
    if cfg.D_norm == 'batch':
        assert any('D/d_conv_3/batch_norm/BatchNorm/gamma' in r for r in checked), sorted(checked)[:20]
    _assert(rows)
    net.set_variables(values)
    net.train_step()
    l = net.losses()
    assert all(np.isfinite(v) for v in l.values()), l
    ev = net.evaluate()
    assert all(np.isfinite(v) for v in ev.values() if np.isscalar(v)), ev
    net.close()
    if len(flags) >= 3:      # the three-norm cases once more in the two other arithmetic modes (the block norms stay fp32 kernels)
        for dtype in ('f32x3', 'bf16'):
            cfg2 = ea.Config(batch_size=3, compute_dtype=dtype, **flags)
            net = ea.EdgeGAN(cfg2, values)
            bf = dtype == 'bf16'       # (bf16 mode: the tolerances of tests/test_bf16_gpu.py against the operand-rounding oracle)
            rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg2, seed=37), bf16=bf, self_consistency=True,
                                         grad_tol=0.1 if bf else 1e-2, loss_tol=1e-2 if bf else 1e-4)
            if bf:      # a PReLU slope's gradient is ONE sum of signed per-element terms (conditioning 30 ... 100 at a random
                        # initialisation, tests/test_tight_gpu.py): bf16 operand rounding reaches 0.10 ... 0.11 on the smallest of them
                rows = [(n, e, max(t, 0.2)) if n.endswith('prelu/param') or 'prelu/param ' in n else (n, e, t) for n, e, t in rows]
            _assert(rows)
            net.close()


@pytest.mark.parametrize('flags', [dict(), dict(E_norm='batch', G_norm='batch'), dict(E_norm=None, G_norm=None)],
                         ids=['default', 'batch_norms', 'no_norms'])
def test_inference_matches_oracle(flags):
    """The test graph (edgegan.py:492-517,591-609) under the three block norms of normalization.py:10-29."""
    cfg = ea.Config(batch_size=2, **flags)
    values = ea.init_variables(cfg, 2)
    rng = np.random.RandomState(6)
    for n in values:      # (move BatchNorm's scale / shift off their initial 1 / 0)
        if n.endswith('BatchNorm/gamma'):
            values[n] = (1.0 + 0.2 * rng.standard_normal(values[n].shape)).astype(np.float32)
        elif n.endswith('BatchNorm/beta'):
            values[n] = (0.1 * rng.standard_normal(values[n].shape)).astype(np.float32)
    net = ea.EdgeGAN(cfg, values)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    for nb in (1, 3):      # the reference tests with batch 1 (test.py:89); BN uses batch statistics
        imgs = np.random.RandomState(nb).uniform(-1, 1, size=(nb, 64, 128, 3)).astype(np.float32)
        cls = (np.arange(nb) % 14).astype(np.int32)
        e_ref, i_ref = om.infer(parity.t64(imgs), torch.tensor(cls), 0.4, -1.1)
        e, i = net.infer(imgs, cls, eps2=[0.4, -1.1])
        assert parity.relerr(e, e_ref.numpy()) < 1e-4 and parity.relerr(i, i_ref.numpy()) < 1e-4
    net.close()


def test_inference_shares_the_encoder_trunk_without_changing_a_bit(monkeypatch):
    """The reference's two runs (E -> G1, E -> G2; edgegan.py:606-609) each evaluate the encoder; the engine evaluates
    the trunk once and applies each run's eps in z = mu + eps exp(log sigma) (model.cpp:infer).  The edge output may
    depend on eps2[0] only, the image output on eps2[1] only, and both must equal -- bit for bit -- what a call that
    gives BOTH runs that eps produces; the two-lane schedule must equal the one-stream one."""
    cfg = ea.Config(batch_size=4)
    values = ea.init_variables(cfg, 5)
    imgs = np.random.RandomState(11).uniform(-1, 1, size=(4, 64, 128, 3)).astype(np.float32)
    cls = np.array([0, 3, 7, 13], dtype=np.int32)
    outs = {}
    for lanes in ('15', '0'):
        monkeypatch.setenv('EG_LANES', lanes)
        net = ea.EdgeGAN(cfg, values)
        outs[lanes] = {k: net.infer(imgs, cls, eps2=list(k)) for k in ((0.4, -1.1), (0.4, 0.4), (-1.1, -1.1), (0.4, 2.0))}
        net.close()
    for lanes, o in outs.items():
        assert np.array_equal(o[(0.4, -1.1)][0], o[(0.4, 0.4)][0]) and np.array_equal(o[(0.4, -1.1)][0], o[(0.4, 2.0)][0])
        assert np.array_equal(o[(0.4, -1.1)][1], o[(-1.1, -1.1)][1])
        assert not np.array_equal(o[(0.4, -1.1)][1], o[(0.4, 2.0)][1])          # eps does reach the image run
    for k in outs['15']:
        assert np.array_equal(outs['15'][k][0], outs['0'][k][0]) and np.array_equal(outs['15'][k][1], outs['0'][k][1])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_inference_calls_in_flight_return_the_same_bits(dtype, monkeypatch):
    """eg_infer_run rotates consecutive calls over slots with their own lanes / arenas / output buffers (capi.cpp): four
    single-chain calls in flight (EG_INFER_SLOTS=4), two calls of two lanes each (2, the default), or one (1: round 5's
    schedule).  Whatever the schedule, and however many runs were enqueued before the download, the outputs are the bits of
    the one-stream schedule; a new upload or a changed variable in between is honoured."""
    cfg = ea.Config(batch_size=8, num_classes=0, compute_dtype=dtype)
    values = ea.init_variables(cfg, 3)
    rng = np.random.RandomState(5)
    a = rng.uniform(-1, 1, size=(8, 64, 128, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(8, 64, 128, 3)).astype(np.float32)
    monkeypatch.setenv('EG_LANES', '0')
    ref = ea.EdgeGAN(cfg, values)
    want_a, want_b = ref.infer(a, None, eps2=[0.3, -0.2]), ref.infer(b, None, eps2=[0.3, -0.2])
    w2 = dict(values)
    w2['G1/g_dconv_4/deconv2d/b'] = values['G1/g_dconv_4/deconv2d/b'] + 0.25
    ref.set_variable('G1/g_dconv_4/deconv2d/b', w2['G1/g_dconv_4/deconv2d/b'])
    want_b2 = ref.infer(b, None, eps2=[0.3, -0.2])
    ref.close()
    monkeypatch.setenv('EG_LANES', '15')
    for slots in ('8', '4', '2', '1'):
        monkeypatch.setenv('EG_INFER_SLOTS', slots)
        net = ea.EdgeGAN(cfg, values)
        net.infer_upload(a, None, eps2=[0.3, -0.2])
        for runs in (1, 2, 3, 9):
            for _ in range(runs):
                net.infer_run()
            e, i = net.infer_download()
            assert np.array_equal(e, want_a[0]) and np.array_equal(i, want_a[1]), (slots, runs)
        for _ in range(5):
            net.infer_run()
        net.infer_upload(b, None, eps2=[0.3, -0.2])       # replaces the batch the runs in flight read: drained first
        for _ in range(6):
            net.infer_run()
        e, i = net.infer_download()
        assert np.array_equal(e, want_b[0]) and np.array_equal(i, want_b[1]), slots
        for _ in range(3):
            net.infer_run()
        net.set_variable('G1/g_dconv_4/deconv2d/b', w2['G1/g_dconv_4/deconv2d/b'])
        for _ in range(5):
            net.infer_run()
        e, i = net.infer_download()
        assert np.array_equal(e, want_b2[0]) and np.array_equal(i, want_b2[1]), slots
        assert not np.array_equal(e, want_b[0])
        net.close()


@pytest.mark.parametrize('flags', [dict(), dict(E_norm='batch', G_norm='batch', D_norm='batch')], ids=['default', 'batch_norms'])
def test_train_step_equals_sequential_substeps_and_is_deterministic(flags):
    """(batch_norms: every variable includes the block norms' beta / gamma, their RMSProp slots and all moving statistics --
    the concurrent schedule advances them exactly as the seven sequential sub-steps do)"""
    cfg = ea.Config(batch_size=2, **flags)
    values = ea.init_variables(cfg, 0)
    batch = ea.synthetic_batch(cfg)
    outs = []
    for mode in ('step', 'substeps', 'step'):
        net = ea.EdgeGAN(cfg, values)
        net.upload_batch(batch['images'], batch['z'], batch['alpha'], batch['eps'])
        if mode == 'step':
            net.train_step()
        else:
            for k in range(1, 8):
                net.substep(k, True)
        outs.append(net.get_variables(slots=True))
        net.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k     # same launches => bitwise equal
        assert np.array_equal(outs[0][k], outs[2][k]), k     # run-to-run deterministic (no float atomics)
    # the dead `disc` head never moves and keeps its slots at their initial values (quirk N8)
    assert np.array_equal(outs[0]['D2/Conv_1/weights'], values['D2/Conv_1/weights'])
    assert np.all(outs[0]['D2/Conv_1/weights/RMSProp'] == 1)


def test_context_recovers_from_an_error_in_the_middle_of_a_backward_sweep(monkeypatch):
    """an exception that unwinds out of a concurrent iteration's backward sweep leaves another lane active, the
    context's stream on that lane's LEAF stream and leaf_dirty set.  The next C-ABI call must start from a sane state
    (csrc/model.cpp: api_enter) -- after the injected fault two iterations must reproduce a fresh engine bit for bit."""
    monkeypatch.setenv('EG_TEST_HOOKS', '1')   # (the fault-injection op exists only where this is set: capi.cpp)
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)
    batch = ea.synthetic_batch(cfg)
    outs = []
    for inject in (False, True):
        net = ea.EdgeGAN(cfg, values)
        net.upload_batch(**batch)
        net.train_step()
        if inject:
            with pytest.raises(ea.EdgeGANLibError, match='injected fault'):
                net.test_op('fault_mid_backward', [], [], [])
            with pytest.raises(ea.EdgeGANLibError, match='injected fault'):      # twice in a row: recovery is idempotent
                net.test_op('fault_mid_backward', [], [], [])
        net.train_step()
        outs.append((net.losses(), net.get_variables(slots=True)))
        net.close()
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def test_pipelined_feed_and_loss_readback_equal_the_synchronous_calls():
    """train.py's default loop is eg_train_batch + eg_losses_wait (update_model_async / losses_async: feed through
    two pinned host slots, losses one iteration late).  It must be the same computation as upload_batch + train_step +
    read_losses: per-iteration losses (both back = 1 and back = 0 slot selection) and the final state, bit for bit."""
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 2)
    batches = [ea.synthetic_batch(cfg, seed=300 + i) for i in range(5)]
    sync = ea.EdgeGAN(cfg, values)
    want = []
    for b in batches:
        sync.update_model(b['images'], b['z'], alpha=b['alpha'], eps=b['eps'])
        want.append(sync.losses())
    want_state = sync.get_variables(slots=True)
    sync.close()
    net = ea.EdgeGAN(cfg, values)
    got = []
    for i, b in enumerate(batches):
        net.update_model_async(b['images'], b['z'], alpha=b['alpha'], eps=b['eps'])
        if i == 2:
            got.append(net.losses_async(back=1))     # iteration 1, asked for AFTER iteration 2 was enqueued ...
            got.append(net.losses_async(back=0))     # ... then iteration 2 at once (the checkpoint-iteration flush of train.py)
        elif i in (1, 4):                            # (i == 3: iteration 2 was already reported at once)
            got.append(net.losses_async(back=1))     # iterations 0 and 3, each one enqueue late
    got.append(net.losses_async(back=0))             # iteration 4: the loop's last read-back
    state = net.get_variables(slots=True)
    net.close()
    assert len(got) == len(want) == 5
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (i, a, b)
    for k in want_state:
        assert np.array_equal(want_state[k], state[k]), k


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_concurrent_lanes_equal_one_stream_over_many_iterations(dtype, monkeypatch):
    """The concurrent sub-step schedule against the one-stream schedule: 25 iterations with a fresh
    batch each must leave every variable and optimizer slot bitwise identical (a race between lanes would not)."""
    import hashlib
    cfg = ea.Config(batch_size=8, compute_dtype=dtype)
    values = ea.init_variables(cfg, 3)
    digests = []
    for lanes in ('15', '0', '7'):              # 15 = the default: sub-steps 1-4 / 6-7 / the critics of 5 and 7 on lanes + leaf streams
        monkeypatch.setenv('EG_LANES', lanes)       # read when the context is created
        net = ea.EdgeGAN(cfg, values)
        for it in range(25):
            b = ea.synthetic_batch(cfg, seed=100 + it)
            net.update_model(b['images'], b['z'], alpha=b['alpha'], eps=b['eps'])
        h = hashlib.sha256()
        for name, v in sorted(net.get_variables(slots=True).items()):
            h.update(name.encode())
            h.update(np.ascontiguousarray(v).tobytes())
        digests.append(h.hexdigest())
        net.close()
    assert digests[0] == digests[1] == digests[2], digests


@pytest.mark.parametrize('dtype', ['f32', 'f32x3'])
def test_branch_lanes_when_the_resize_is_the_identity(dtype, monkeypatch):
    """dis_size == H == W/2: tf.image.resize_images returns its input, so in sub-steps 5 / 7 the image
    discriminator and the classifier both differentiate w.r.t. the SAME tensor G2(z).  With the critics on their own
    lanes (EG_LANES bit 2) the image discriminator's accumulation into that gradient must be ordered behind the
    classifier's first write (model.cpp:run_substep): 6 iterations with lanes on / off / on must agree bit
    for bit, for the whole state."""
    import hashlib
    cfg = ea.Config(batch_size=4, dis_size=64, compute_dtype=dtype)
    values = ea.init_variables(cfg, 9)
    digests = []
    for lanes in ('15', '0', '7'):
        monkeypatch.setenv('EG_LANES', lanes)
        net = ea.EdgeGAN(cfg, values)
        for it in range(6):
            b = ea.synthetic_batch(cfg, seed=40 + it)
            net.update_model(b['images'], b['z'], alpha=b['alpha'], eps=b['eps'])
        h = hashlib.sha256()
        for name, v in sorted(net.get_variables(slots=True).items()):
            assert np.all(np.isfinite(v)), name
            h.update(name.encode())
            h.update(np.ascontiguousarray(v).tobytes())
        digests.append(h.hexdigest())
        net.close()
    assert digests[0] == digests[1] == digests[2], digests


@pytest.mark.parametrize('dtype', ['f32x3', 'bf16'])
def test_gradient_probes_reproduce_themselves_and_the_one_stream_schedule(dtype, monkeypatch):
    """eg_substep(k, apply=0) -- the path every parity test reads gradients through -- runs ONE backward sweep on the
    lane's main stream with the filter gradients and the classifier's image-pyramid gradient chain on its leaf stream
    (ops.cpp: LeafScope).  Everything the main stream reads or accumulates behind a leaf-stream writer must be ordered
    behind it (round 3 found d(G2(z)) read by the resize / concat / generator backward before the pyramid chain had
    finished): each sub-step's gradients must come out bit-identical when the probe is repeated, and identical to the
    one-stream schedule (EG_LANES=0)."""
    cfg = ea.Config(batch_size=16, compute_dtype=dtype)
    values = ea.init_variables(cfg, 2)
    b = ea.synthetic_batch(cfg, seed=17)
    names = {1: ['D/d_conv_1/conv2d/w'], 4: ['D2/mru_conv_unit_t_2_layer_0/Conv_2/weights', 'D2/Conv/weights'],
             5: ['G2/g_dconv_3/deconv2d/w', 'G2/g_lin_0/Matrix', 'G1/g_dconv_1/deconv2d/w'], 6: ['E/e_resnet_128_1/res1/conv2d/w'],
             7: ['G2/g_dconv_4/deconv2d/w', 'G1/g_dconv_3/deconv2d/w']}
    got = {}
    for lanes in ('15', '0'):
        monkeypatch.setenv('EG_LANES', lanes)
        net = ea.EdgeGAN(cfg, values)
        net.upload_batch(**b)
        for rep in range(3):
            for k, vns in names.items():
                net.substep(k, apply=False)
                for vn in vns:
                    g = net.get_variable(vn + '/grad')
                    assert np.all(np.isfinite(g)), (lanes, rep, k, vn)
                    key = (k, vn)
                    if key in got:
                        assert np.array_equal(g, got[key]), f'EG_LANES={lanes} repetition {rep}: sub-step {k} {vn} differs'
                    else:
                        got[key] = g
        net.close()


def test_rccl_single_rank_allreduce_and_overlap_path():
    """nranks=1 communicator: exercises the RCCL binding, the all-reduce on the real library and the side-stream
    overlap of (all-reduce + RMSProp) with the next sub-step's compute.  With one rank the exchange is the
    identity, so two full iterations must reproduce the communicator-free run bit for bit."""
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)
    batch = ea.synthetic_batch(cfg)
    outs = []
    for use_comm in (False, True):
        net = ea.EdgeGAN(cfg, values)
        if use_comm:
            net.comm_init(net.comm_unique_id(), 0, 1)
        net.upload_batch(**batch)
        net.train_step()
        net.train_step()
        outs.append((net.get_variables(slots=True), net.losses()))
        net.close()
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert outs[0][1] == outs[1][1]


def test_exchange_footprint_knobs_do_not_change_a_bit(monkeypatch):
    """EG_COMM_CUS (the communication stream confined to n compute units by a CU mask) and EG_COMM_CHANNELS (RCCL's channel cap,
    exported as NCCL_MAX_NCHANNELS) are placement knobs (csrc/comm.cpp): with both set, a one-rank data-parallel run on the real
    RCCL -- masked communication stream, all-reduce + averaging RMSProp on it, per-collective timing on -- reproduces the plain run
    bit for bit, and the seven collectives of an iteration are recorded.  (What the knobs do to an 8-device exchange cannot be
    measured on this pool; a complementary mask on the compute lanes was measured to cost 29 % and removed.)"""
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)
    batch = ea.synthetic_batch(cfg)
    ref = ea.EdgeGAN(cfg, values)
    ref.upload_batch(**batch)
    ref.train_step()
    want = ref.get_variables(slots=True)
    ref.close()
    for k, v in (('EG_COMM_CUS', '16'), ('EG_COMM_CHANNELS', '4')):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv('NCCL_MAX_NCHANNELS', raising=False)
    net = ea.EdgeGAN(cfg, values)
    net.comm_init(net.comm_unique_id(), 0, 1)
    import ctypes
    getenv = ctypes.CDLL(None).getenv          # (the library's setenv is not visible through os.environ's snapshot)
    getenv.restype = ctypes.c_char_p
    assert getenv(b'NCCL_MAX_NCHANNELS') == b'4'
    net.comm_timing(True)
    net.upload_batch(**batch)
    net.train_step()
    net.sync()
    recs = net.comm_timing_read()
    got = net.get_variables(slots=True)
    net.close()
    assert len(recs) == 7 and sorted(g for g, _, _ in recs) == sorted(['D', 'D_patch2', 'D_patch3', 'D2', 'G1+G2', 'E', 'G1+G2'])
    assert all(ms >= 0.0 and nbytes > 0 for _, nbytes, ms in recs)
    for k in want:
        assert np.array_equal(want[k], got[k]), k


def test_checkpoint_round_trip_on_device(tmp_path):
    """save -> fresh engine -> load: every variable and RMSProp slot identical; resume counter from the name."""
    from edgegan_amd import checkpoint, tf_bundle
    cfg = ea.Config(batch_size=2)
    net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 3))
    net.upload_batch(**ea.synthetic_batch(cfg))
    net.train_step()
    d = str(tmp_path / 'checkpoints')
    checkpoint.save(net, d, 502)
    want = net.get_variables(slots=True)
    net.close()
    keys = tf_bundle.list_bundle(d + '/EdgeGAN-Model-502')
    assert len(keys) == 159 + 2 * (sum(1 for s in ea.variable_specs(cfg) if s.trainable) - 2)
    fresh = ea.EdgeGAN(cfg)
    ok, counter = checkpoint.load(fresh, d, strict=True)
    assert ok and counter == 502
    got = fresh.get_variables(slots=True)
    for k, v in want.items():
        if k.startswith('D2/Conv_1/') and 'RMSProp' in k:
            continue
        assert np.array_equal(got[k], v), k
    fresh.close()


def test_train_and_test_command_lines_end_to_end(tmp_path, monkeypatch):
    """python -m edgegan_amd.train / .test on a tiny generated dataset: output tree, flags.json, checkpoint at
    counter % frequency == 2, resume, generated PNGs (edgegan/train.py, edgegan/test.py, edgegan.py:425-633)."""
    from PIL import Image
    from edgegan_amd import train as TR, test as TE, tf_bundle
    rng = np.random.RandomState(0)
    root = tmp_path / 'data'
    for c in (0, 5):
        for phase in ('train', 'test'):
            os.makedirs(root / 'class14' / phase / str(c))
            for i in range(2):
                Image.fromarray(rng.randint(0, 256, size=(64, 128, 3)).astype(np.uint8)).save(
                    root / 'class14' / phase / str(c) / f'{i}.png')
    monkeypatch.chdir(tmp_path)
    args = ['--name=t1', f'--dataroot={root}', '--batch_size=2', '--epoch=1', '--save_checkpoint_frequency=3', '--noreference_evaluation']
    TR.main(args)
    out = tmp_path / 'outputs' / 't1'
    assert (out / 'flags.json').exists() and (out / 'logs').is_dir()
    # counter starts at 1, +1 per iteration; saved when counter % 3 == 2 -> after the 1st iteration (counter 2)
    assert (out / 'checkpoints' / 'EdgeGAN-Model-2.index').exists()
    assert tf_bundle.read_checkpoint_state(str(out / 'checkpoints'))[0] == 'EdgeGAN-Model-2'
    from edgegan_amd import summary
    ev = summary.read_events(str(next((out / 'logs').glob('events.out.tfevents.*'))))
    assert len(ev) == 1 + 2 and set(ev[1][1]) == set(summary.SCALAR_TAGS) and ev[1][0] == 1 and ev[2][0] == 2
    # the reference's own read-back semantics: post-update evaluation + histogram / image summaries (edgegan.py:344-411)
    # (the DEFAULT since round 6: a user of the reference sees the reference's numbers)
    TR.main(['--name=t2', f'--dataroot={root}', '--batch_size=2', '--epoch=1'])
    tags = summary.read_event_tags(str(next((tmp_path / 'outputs' / 't2' / 'logs').glob('events.out.tfevents.*'))))
    kinds = {}
    for _, t in tags:
        kinds.update(t)
    assert kinds['d'] == kinds['edgeDfake'] == kinds['z'] == 'histo' and kinds['G1/image/0'] == kinds['inputs/image/1'] == 'image'
    assert kinds['resized_G_p3_image/image/0'] == 'image' and kinds['joint_dis_dloss'] == 'scalar'
    TE.main(['--name=t1', f'--dataroot={root}'])
    pngs = sorted((out / 'test_output' / 'class14').rglob('*.png'))
    assert len(pngs) == 4
    assert Image.open(pngs[0]).size == (256, 64)          # 'full': input | edge | image  (edgegan.py:617-619)


def test_high_resolution_config_forwards():
    """configs[4] geometry (256x512, 14 classes): 5-block encoder (encoder.py:57-58), 256->128 legacy-bicubic
    (= 2x subsampling) into the patch discriminators, 131072-wide generator projection."""
    cfg = ea.Config(batch_size=1, height=256, width=512)
    values = ea.init_variables(cfg, 4)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.net_checks(net, values)
    _assert(rows)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    imgs = np.random.RandomState(0).uniform(-1, 1, size=(1, 256, 512, 3)).astype(np.float32)
    e_ref, i_ref = om.infer(parity.t64(imgs), torch.tensor([5]), 0.1, 0.2)
    e, i = net.infer(imgs, np.array([5], np.int32), eps2=[0.1, 0.2])
    assert parity.relerr(e, e_ref.numpy()) < 1e-4 and parity.relerr(i, i_ref.numpy()) < 1e-4
    net.close()


def test_substeps_high_resolution_config():
    """configs[4] geometry as a TRAINING step: 14-class 256x512, batch 1 -- all 7 sub-steps (gradients incl. the
    WGAN-GP double backward through the 2x-decimating legacy bicubic into the patch discriminators, the 5-block
    encoder, the 131072-wide generator projection) against oracle autograd in fp64."""
    cfg = ea.Config(batch_size=1, height=256, width=512)
    values = ea.init_variables(cfg, 7)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=5))
    assert len(rows) > 400
    _assert(rows)
    net.close()


def test_full_size_classifier_step_equals_mean_of_small_shards():
    """Size-independent property at BASELINE's full size (batch 64, 14 classes): sub-step 4's loss is a batch
    mean over per-sample-independent computations (no BatchNorm on this path), so its gradient at B=64 must equal
    the mean of the gradients of 32 shards of 2 -- the small-batch path is the one pinned against the oracle, the
    B=64 path exercises the 128x128 tiles, split wgrads and the thin-layer kernels at production shapes.

    Tolerance: the two evaluations run different kernels (tile shapes, summation orders), i.e. carry different fp32
    rounding, and this network amplifies rounding through its min-max gates and PReLU kinks (DESIGN.md section 4,
    "Conditioning").  Measured: 1e-6 ... 9e-4 on the filters, 1.4e-3 on the scalar PReLU slope (a sum of 8.4 M signed
    terms that cancels to 1e-4); with the round-1 thin-layer kernels, whose op-level error against the fp64 oracle is
    3-8x LARGER (profiles/r02c_thin_op_errors.txt), the same quantities were 1e-6 ... 3.5e-4 and 5e-4.  A wiring or
    indexing error shows up as >= 1e-1."""
    names = ['D2/Conv/weights', 'D2/mru_conv_unit_t_1_layer_0/Conv_2/weights', 'D2/mru_conv_unit_t_1_layer_0/update_gate/weights',
             'D2/mru_conv_unit_t_2_layer_0/Conv_1/weights', 'D2/mru_conv_unit_t_4_layer_0/Conv_2/weights',
             'D2/mru_conv_unit_t_3_layer_0/Conv_3/biases', 'D2/mru_conv_unit_t_2_layer_0/norm_activation_in/prelu/param',
             'D2/fully_connected/weights']
    big_cfg, small_cfg = ea.Config(batch_size=64), ea.Config(batch_size=2)
    values = ea.init_variables(big_cfg, 0)
    batch = ea.synthetic_batch(big_cfg)
    big = ea.EdgeGAN(big_cfg, values)
    big.upload_batch(**batch)
    big.substep(4, apply=False)
    want = {n: big.get_variable(n + '/grad').astype(np.float64) for n in names}
    loss_big = big.losses()['loss_d_ac']
    big.close()
    small = ea.EdgeGAN(small_cfg, values)
    acc = {n: 0.0 for n in names}
    loss = 0.0
    for s in range(32):
        sl = slice(2 * s, 2 * s + 2)
        small.upload_batch(batch['images'][sl], batch['z'][sl], batch['alpha'][:, sl].copy(), batch['eps'])
        small.substep(4, apply=False)
        for n in names:
            acc[n] = acc[n] + small.get_variable(n + '/grad').astype(np.float64) / 32
        loss += small.losses()['loss_d_ac'] / 32
    small.close()
    assert abs(loss - loss_big) / abs(loss_big) < 1e-5
    for n in names:
        assert parity.rel_l2(want[n], acc[n]) < 4e-3, (n, parity.rel_l2(want[n], acc[n]))


def test_inference_on_reference_example_sketches():
    """SURVEY 8c golden item (iv): E -> G1, G2 on the reference's 4 example test PNGs (loaded through the data
    pipeline, class from the directory name, fixed eps), batch 1 like edgegan.test, against the oracle."""
    from edgegan_amd import data as D
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    dcfg = {'input_height': 64, 'input_width': 128, 'output_height': 64, 'output_width': 128, 'crop': False}
    ds = D.Dataset(golden, 'dataset_example', np.inf, 1, dcfg, None, 'test')
    cfg = ea.Config(batch_size=1)
    values = ea.init_variables(cfg, 7)
    net = ea.EdgeGAN(cfg, values)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    assert len(ds) == 4
    for idx in range(len(ds)):
        imgs, files = ds[idx]
        cls = np.array([int(os.path.basename(os.path.dirname(files[0])))], dtype=np.int32)
        e_ref, i_ref = om.infer(parity.t64(imgs), torch.tensor(cls), 0.3, -0.7)
        e, i = net.infer(imgs, cls, eps2=[0.3, -0.7])
        assert parity.relerr(e, e_ref.numpy()) < 1e-4 and parity.relerr(i, i_ref.numpy()) < 1e-4, files[0]
    net.close()


def test_post_update_evaluation_matches_oracle_forward_graph():
    """eg_eval_losses = the reference's observe-only runs after update_model (edgegan.py:427-433,464-480): forward-only
    with the current weights, caller-supplied alpha / eps, BatchNorm moving statistics advanced bn_updates times."""
    cfg = ea.Config(batch_size=3)
    values = ea.init_variables(cfg, 4)
    b = ea.synthetic_batch(cfg, seed=21)
    net = ea.EdgeGAN(cfg, values)
    net.upload_batch(**b)
    net.train_step()                      # evaluate with POST-update weights
    cur = net.get_variables()
    rng = np.random.RandomState(1)
    alpha, eps = rng.uniform(size=(3, 3)).astype(np.float32), 0.7
    ev = net.evaluate(alpha=alpha, eps=eps, bn_updates=6, tensors=True)
    om = O.OracleModel(parity.oracle_config(cfg), cur, dtype=torch.float64)
    inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(alpha), parity.t64([eps]))
    bn = {}
    ref = O.forward_graph(om.p, inp, om.cfg, eps, bn_stats=bn)
    for k in ('joint_dis_dloss', 'image_dis_dloss', 'edge_dis_dloss', 'loss_d_ac', 'edge_gloss', 'image_gloss', 'zl_loss',
              'joint_dis_gloss', 'edge_dis_gloss', 'image_dis_gloss', 'loss_g_ac'):
        assert abs(ev[k] - float(ref[k])) <= 2e-4 * abs(float(ref[k])) + 1e-6, (k, ev[k], float(ref[k]))
    # discriminator logits behind the "d" / "d_" / "imageD" ... histograms
    W2 = cfg.width // 2
    edge, image = O.generator(om.p, 'G1', O.z_onehot(inp.z, om.cfg)[0], om.cfg), O.generator(om.p, 'G2', O.z_onehot(inp.z, om.cfg)[0], om.cfg)
    assert parity.relerr(ev['edge_output'], edge.detach().numpy()) < 1e-4
    assert parity.relerr(ev['image_output'], image.detach().numpy()) < 1e-4
    d_real = O.discriminator(om.p, 'D', inp.images)[1].detach().numpy().reshape(-1)
    d_fake = O.discriminator(om.p, 'D', torch.cat([edge, image], dim=2))[1].detach().numpy().reshape(-1)
    assert parity.relerr(ev['d_logits'][0], d_real) < 1e-3 and parity.relerr(ev['d_logits'][1], d_fake) < 1e-3
    r_edges = O.resize_bicubic(inp.images[:, :, :W2, :], cfg.dis_size).numpy()
    assert parity.relerr(ev['resized_edges'], r_edges) < 1e-5
    # BatchNorm moving statistics advanced 6 times with the (identical) batch statistics of this forward
    for g in ('G1', 'G2'):
        bm, bv, n = bn[g]
        for name, stat in (('moving_mean', bm), ('moving_variance', bv * (n / (n - 1)))):
            vn = f'{g}/batch_norm/BatchNorm/{name}'
            want = torch.tensor(cur[vn], dtype=torch.float64)
            for _ in range(6):
                want = 0.9 * want + 0.1 * stat
            assert parity.relerr(net.get_variable(vn), want.numpy()) < 1e-4, vn
    # the weights themselves are untouched by an evaluation
    after = net.get_variables()
    for k, v in cur.items():
        if 'moving_' not in k:
            assert np.array_equal(v, after[k]), k
    net.close()


def test_mfma_probe_reports_the_matrix_pipes_sustained_rate():
    """eg_mfma_probe (bench.py's roofline context): a memory-free bf16 MFMA loop cannot beat the nominal dense peak, and
    random operands cannot be faster than constant ones (they only cost clock)."""
    net = ea.EdgeGAN(ea.Config(batch_size=2))
    const, rnd = net.mfma_probe(4000, False), net.mfma_probe(4000, True)
    net.close()
    assert 500.0 < rnd <= const * 1.03 and const < 2600.0, (const, rnd)


def test_engine_create_destroy_returns_its_device_memory():
    """eg_create ... eg_destroy in a loop (a training job, an evaluation job, a restart in one process): every buffer, stream and
    event of a context is released with it.  Free device memory (hipMemGetInfo) after the second and after the sixth cycle --
    training iteration, evaluation and inference in each, all three arithmetic modes and the block-norm variants -- must agree."""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    free, total = ctypes.c_size_t(), ctypes.c_size_t()

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    kinds = [dict(), dict(compute_dtype='bf16'), dict(compute_dtype='f32x3'), dict(E_norm='batch', G_norm='batch', D_norm='batch'),
             dict(D_norm=None), dict(compute_dtype='bf16', num_classes=0)]
    seen = []
    for rep in range(3):
        for kw in kinds:
            cfg = ea.Config(batch_size=8, **kw)
            net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 1))
            b = ea.synthetic_batch(cfg, seed=3)
            net.upload_batch(**b)
            net.train_step()
            assert all(np.isfinite(v) for v in net.losses().values())
            net.evaluate()
            net.infer(b['images'], None if not cfg.multiclasses else (np.arange(8) % cfg.num_classes).astype(np.int32), eps2=[0.1, 0.2])
            net.close()
        seen.append(free_bytes())
    print('free device memory after each pass over the six configurations:', seen)
    assert abs(seen[2] - seen[1]) < (32 << 20), seen      # (the first pass may leave the runtime's own pools / code objects behind)
