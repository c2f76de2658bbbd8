"""GPU parity of the networks, the 7 optimizer sub-steps (gradients incl. the WGAN-GP double
backward, losses, RMSProp application, BatchNorm moving statistics) and inference."""
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
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg))
    assert len(rows) > 400
    _assert(rows)
    net.close()


def test_substeps_single_class_batch4():
    """config[0]: single-class 64x128, batch 4 (edgegan.train --nomulticlasses)."""
    cfg = ea.Config(batch_size=4, num_classes=0)
    values = ea.init_variables(cfg, 5)
    net = ea.EdgeGAN(cfg, values)
    _assert(parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=99)))
    net.close()


def test_inference_matches_oracle():
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 2)
    net = ea.EdgeGAN(cfg, values)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    for nb in (1, 3):      # the reference tests with batch 1 (test.py:89); BN uses batch statistics
        imgs = np.random.RandomState(nb).uniform(-1, 1, size=(nb, 64, 128, 3)).astype(np.float32)
        cls = (np.arange(nb) % 14).astype(np.int32)
        e_ref, i_ref = om.infer(parity.t64(imgs), torch.tensor(cls), 0.4, -1.1)
        e, i = net.infer(imgs, cls, eps2=[0.4, -1.1])
        assert parity.relerr(e, e_ref.numpy()) < 1e-4 and parity.relerr(i, i_ref.numpy()) < 1e-4
    net.close()


def test_train_step_equals_sequential_substeps_and_is_deterministic():
    cfg = ea.Config(batch_size=2)
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


def test_rccl_single_rank_allreduce_path():
    """nranks=1 communicator: exercises RCCL binding + the all-reduce call on the real library."""
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)
    batch = ea.synthetic_batch(cfg)
    ref = ea.EdgeGAN(cfg, values)
    ref.upload_batch(**batch)
    ref.substep(6, True)
    want = ref.get_variable('E/FC8_mu/w')
    ref.close()
    net = ea.EdgeGAN(cfg, values)
    net.comm_init(net.comm_unique_id(), 0, 1)
    net.upload_batch(**batch)
    net.substep(6, True)
    assert np.array_equal(net.get_variable('E/FC8_mu/w'), want)
    net.close()
