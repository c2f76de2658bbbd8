"""CPU-side checks: the C-ABI library builds, loads and exports every symbol of include/edgegan_hip.h;
the variable registry matches the oracle's restatement; the engine refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import edgegan_amd as ea
from edgegan_amd import _lib
import edgegan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, 'include', 'edgegan_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(eg_[a-z_0-9]+)\s*\(', header)))
    assert declared, 'no declarations parsed'
    missing = [s for s in declared if not hasattr(built_lib, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared


@pytest.mark.parametrize('num_classes', [14, 0])
def test_registry_matches_oracle_restatement(built_lib, num_classes):
    cfg = ea.Config(batch_size=4, num_classes=num_classes)
    got = [(s.name, tuple(s.shape), s.trainable) for s in ea.variable_specs(cfg)]
    ocfg = O.Config(batch_size=4, num_classes=num_classes)
    want = [(s.name, tuple(s.shape), s.trainable) for s in O.variable_specs(ocfg)]
    assert got == want
    n_train = sum(int(np.prod(s[1], dtype=np.int64)) for s in got if s[2])
    # 45.9 M trainable parameters with 14 classes
    if num_classes == 14:
        assert n_train == 2 * 5249155 + 2771969 + 2 * 2788353 + 9907592 + 17151293


@pytest.mark.parametrize('flags', [dict(use_image_discriminator=False), dict(use_edge_discriminator=False),
                                   dict(image_dis_size=64, edge_dis_size=256)])
def test_registry_non_default_discriminator_flags(built_lib, flags):
    """edgegan/train.py:61-67: a patch discriminator that is switched off is never constructed (models/edgegan.py:164-174);
    each keeps its own input size -- the d_linear_5 matrix follows it."""
    cfg = ea.Config(batch_size=2, **flags)
    got = [(s.name, tuple(s.shape), s.trainable) for s in ea.variable_specs(cfg)]
    want = [(s.name, tuple(s.shape), s.trainable) for s in O.variable_specs(O.Config(batch_size=2, **flags))]
    assert got == want
    names = dict((n, sh) for n, sh, _ in got)
    assert ('D_patch2/d_conv_0/conv2d/w' in names) == cfg.use_image_discriminator
    assert ('D_patch3/d_conv_0/conv2d/w' in names) == cfg.use_edge_discriminator
    if 'image_dis_size' in flags:
        assert names['D_patch2/d_linear_5/Matrix'] == (4 * 4 * 512, 1) and names['D_patch3/d_linear_5/Matrix'] == (16 * 16 * 512, 1)


@pytest.mark.parametrize('flags', [dict(G_norm='batch'), dict(E_norm='batch'), dict(D_norm='batch'),
                                   dict(E_norm='batch', G_norm='batch', D_norm='batch', num_classes=0),
                                   dict(E_norm=None, G_norm=None, D_norm=None)])
def test_registry_batch_norm_flags(built_lib, flags):
    """edgegan/train.py:54-59 E_norm / G_norm / D_norm = 'batch' (nn/modules/normalization.py:19-25): every normalised block
    gains contrib batch_norm's four variables under <block>/batch_norm/BatchNorm -- the generators' g_dconv_1..3
    (generator.py:54-68), the discriminators' d_conv_1, d_conv_3, d_conv_4 (discriminator.py:63-72; d_conv_0 has norm=None),
    both halves of every residual block of the encoder (encoder.py:67, conv.py:73-80); beta / gamma trainable.  None
    (normalization.py:26-27) adds nothing: the registry of the InstanceNorm graph."""
    cfg = ea.Config(batch_size=2, **flags)
    got = [(s.name, tuple(s.shape), s.trainable) for s in ea.variable_specs(cfg)]
    want = [(s.name, tuple(s.shape), s.trainable) for s in O.variable_specs(O.Config(batch_size=2, **flags))]
    assert got == want
    names = {n: (sh, tr) for n, sh, tr in got}
    assert ('G1/g_dconv_2/batch_norm/BatchNorm/gamma' in names) == (cfg.G_norm == 'batch')
    assert ('D_patch2/d_conv_3/batch_norm/BatchNorm/beta' in names) == (cfg.D_norm == 'batch')
    assert 'D/d_conv_0/batch_norm/BatchNorm/beta' not in names and 'G1/g_dconv_4/batch_norm/BatchNorm/beta' not in names
    assert ('E/e_resnet_256_2/res2/batch_norm/BatchNorm/moving_variance' in names) == (cfg.E_norm == 'batch')
    if cfg.G_norm == 'batch':
        assert names['G2/g_dconv_1/batch_norm/BatchNorm/gamma'] == ((256,), True)
        assert names['G2/g_dconv_1/batch_norm/BatchNorm/moving_mean'] == ((256,), False)
    with pytest.raises(KeyError):
        ea.Config(G_norm='layer').to_c()


def test_config_struct_is_versioned(built_lib):
    """eg_config starts with struct_size (include/edgegan_hip.h): eg_default_config writes the library's sizeof, the Python
    binding's ctypes layout has the same size, and a caller with another layout is refused instead of being read shifted."""
    cc = _lib._CConfig()
    assert built_lib.eg_default_config(ctypes.byref(cc)) == 0
    assert cc.struct_size == ctypes.sizeof(_lib._CConfig)
    assert (cc.batch_size, cc.height, cc.width, cc.num_classes, cc.use_image_discriminator, cc.image_dis_size,
            cc.use_edge_discriminator, cc.edge_dis_size, cc.compute_dtype) == (64, 64, 128, 14, 1, 128, 1, 128, 0)
    assert abs(cc.learning_rate - 2e-4) < 1e-9 and cc.lambda_gp == 10.0
    assert built_lib.eg_variable_count(ctypes.byref(cc)) == len(ea.variable_specs(ea.Config()))
    stale = ea.Config().to_c()
    stale.struct_size -= 4                      # a binding built against a header with one field less
    assert built_lib.eg_variable_count(ctypes.byref(stale)) == -1
    ctx = ctypes.c_void_p()
    rc = built_lib.eg_create(ctypes.byref(stale), ctypes.byref(ctx))
    assert rc != 0 and not ctx.value
    assert 'layout mismatch' in built_lib.eg_last_error(None).decode()


def test_registry_high_res_config(built_lib):
    cfg = ea.Config(batch_size=2, height=256, width=512)
    names = {s.name: s.shape for s in ea.variable_specs(cfg)}
    assert names['G1/g_lin_0/Matrix'] == (114, 131072)
    assert names['D/d_linear_5/Matrix'] == (262144, 1)
    assert 'E/e_resnet_512_5/res1/conv2d/w' in names              # encoder.py:57-58


def test_initialisers_follow_reference(built_lib):
    cfg = ea.Config(batch_size=2)
    v = ea.init_variables(cfg, seed=3)
    w = v['D/d_conv_1/conv2d/w']
    assert abs(w.std() - 0.02 * 0.8796) < 2e-3 and np.abs(w).max() <= 0.04 + 1e-7    # truncated at 2 sigma
    assert abs(v['G1/g_dconv_1/deconv2d/w'].std() - 0.02) < 1e-3
    assert np.all(v['D2/mru_conv_unit_t_2_layer_0/update_gate/biases'] == 0.5)
    assert v['D2/Conv/prelu/param'].shape == () and float(v['D2/Conv/prelu/param']) == np.float32(0.2)
    assert np.all(v['G2/batch_norm/BatchNorm/gamma'] == 1) and np.all(v['G2/batch_norm/BatchNorm/moving_variance'] == 1)
    lim = np.sqrt(6.0 / (768 + 14))
    assert np.abs(v['D2/fully_connected/weights']).max() <= lim


def test_no_cpu_fallback(built_lib):
    """Without a GPU the product must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(ea.EdgeGANLibError, match='no CPU fallback'):
        ea.EdgeGAN(ea.Config(batch_size=2))


def test_synthetic_batch_contract():
    cfg = ea.Config(batch_size=8)
    b = ea.synthetic_batch(cfg)
    assert b['images'].shape == (8, 64, 128, 3) and b['images'].dtype == np.float32
    assert b['images'].min() >= -1 and b['images'].max() <= 1
    assert b['z'].shape == (8, 101) and np.all(b['z'][:, -1] == np.arange(8) % 14)
    assert b['alpha'].shape == (3, 8) and 0 <= b['alpha'].min() and b['alpha'].max() <= 1
    b2 = ea.synthetic_batch(cfg)
    assert all(np.array_equal(b[k], b2[k]) for k in b)
    assert not np.array_equal(ea.synthetic_batch(cfg, rank=1)['z'], b['z'])


def test_loopback_rccl_stand_in_builds_and_exports_what_the_engine_binds(built_lib):
    """tests/shim/loopback_rccl.cpp (what tests/test_dp_gpu.py binds through EG_RCCL_LIB) must offer every RCCL entry
    point edgegan_amd/csrc/comm.cpp resolves by name -- checked against the dlsym strings in comm.cpp itself."""
    import __graft_entry__ as ge
    so = os.path.join(ROOT, 'tests', 'shim', 'libloopback_rccl.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, 'tests', 'shim', 'loopback_rccl.cpp')):
        ge.build_test_shim()
    bound = sorted(set(re.findall(r'"(nccl[A-Za-z]+)"', open(os.path.join(ROOT, 'edgegan_amd', 'csrc', 'comm.cpp')).read())))
    assert bound == ['ncclAllReduce', 'ncclCommDestroy', 'ncclCommInitRank', 'ncclGetErrorString', 'ncclGetUniqueId',
                     'ncclGetVersion'], bound
    lib = ctypes.CDLL(so)
    assert all(hasattr(lib, s) for s in bound)
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    assert b'invalid usage' in lib.ncclGetErrorString(5)
