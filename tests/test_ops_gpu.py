"""GPU parity of every tape op (forward + backward through the C ABI) against the oracle."""
import numpy as np
import pytest

import edgegan_amd as ea
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def net():
    n = ea.EdgeGAN(ea.Config(batch_size=2))
    yield n
    n.close()


def _assert(rows):
    bad = [r for r in rows if not r[1] <= r[2]]
    assert not bad, '\n' + parity.format_rows(bad)


@pytest.mark.parametrize('case', parity.CONV_CASES, ids=[c[0] for c in parity.CONV_CASES])
def test_conv2d(net, case):
    _assert(parity.check_conv(net, case))


@pytest.mark.parametrize('case', parity.DECONV_CASES, ids=[c[0] for c in parity.DECONV_CASES])
def test_deconv2d(net, case):
    _assert(parity.check_deconv(net, case))


@pytest.mark.parametrize('case', parity.LONG_M_CASES, ids=[c[0] for c in parity.LONG_M_CASES])
def test_conv2d_many_pixels_per_wgrad_block(net, case):
    """Long pixel runs per weight-gradient block (deep split reductions, > 16 K steps per block)."""
    _assert(parity.check_conv(net, case, seed=31))


@pytest.mark.parametrize('act', [0, 1, 2])
def test_instance_norm_act(net, act):
    _assert(parity.check_in_act(net, act))
    _assert(parity.check_in_act(net, act, shape=(2, 4, 8, 512)))


def test_instance_norm_tangent_and_its_reverse(net):
    _assert(parity.check_in_tangent(net, 2))
    _assert(parity.check_in_tangent(net, 2, shape=(2, 4, 8, 128), seed=33))
    for i, shape in enumerate(parity.TANGENT_SHAPES):
        _assert(parity.check_in_tangent(net, 2, shape=shape, seed=40 + i))


def test_remaining_ops(net):
    rows = [r for r in parity.all_op_checks(net) if not r[0].startswith(('conv2d', 'deconv2d', 'in_'))]
    assert len(rows) > 30
    _assert(rows)


def test_fused_mru_chains(net):
    """conv + PReLU, the update gate, pool2_add + PReLU and the activation -> convolution chains as the classifier / the
    discriminators build them (round 5: second epilogue stores, K-split slabs summed by their consumer)."""
    _assert(parity.fused_chain_checks(net))


def test_bad_arguments_fail_loudly(net):
    with pytest.raises(ea.EdgeGANLibError):
        net.test_op('no_such_op', [np.zeros(4, np.float32)], [1], [(4,)])
    with pytest.raises(ea.EdgeGANLibError):
        net.get_variable('not/a/variable', (1,))
    with pytest.raises(ea.EdgeGANLibError):
        net.set_variable('D/d_linear_5/bias', np.zeros(3, np.float32))
