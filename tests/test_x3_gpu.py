"""compute_dtype='f32x3': fp32 contractions formed from exact three-limb bf16 operand splits on the bf16 MFMA
(edgegan_amd/csrc/k_gemm_x3.hip) -- the mode `bench.py` measures by default.  It is an fp32 implementation, so it is
held to the fp32 tests' OWN tolerances against the fp64 oracle (no emulation in the oracle, unlike the bf16 mode):

  * every conv / deconv op case, forward + input gradient + filter gradient, at the fp32 max-norm bounds 2e-5 / 5e-5
    (measured <= 1.5e-6, the exact-fp32 MFMA kernels measure <= 1e-6 on the same cases);
  * network forwards at 5e-4, inference at 1e-4, the committed golden vectors;
  * the margin-selected fixtures (no activation within 3e-6 rms of a ReLU / PReLU kink) at relative L2 1e-4 on every
    optimised variable of every sub-step, and BASELINE's full batch (B=64) sub-steps 1, 3, 4, 5, 6 against fp64 autograd --
    the tests of tests/test_tight_gpu.py, re-run here with the fp32 compute type mapped to f32x3 (EG_F32_MODE);
  * the ordinary sub-step tests at their relative-L2 1e-2 (kink flips allowed) tolerance.

A different fp32 rounding lands a pre-activation that sits within fp32 rounding of a kink on the other side as often as
not (tests/parity.py:_substep_checks); the two fp32-mode tests that are sensitive to that at their own tolerance
(test_step_gpu.py::test_substeps_odd_batch: a PReLU slope whose gradient is 1e-4 of the sub-step's largest, behind a
pre-activation at 1.4e-7 rms in the oracle; the batch-64-vs-32-shards property at 1e-3) are run here with the flip
accounted for: see the two tests at the end."""
import numpy as np
import pytest

import edgegan_amd as ea
import parity
import test_golden_gpu
import test_step_gpu
import test_tight_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f32_is_x3(monkeypatch):
    monkeypatch.setenv('EG_F32_MODE', 'f32x3')


def _assert(rows):
    bad = [r for r in rows if not r[1] <= r[2]]
    assert not bad, '\n' + parity.format_rows(bad)


def test_x3_fused_mru_chains():
    """The fused forms of the MRU unit in the benched mode: Conv_1's PReLU from the halo kernel's epilogue, K-split slabs
    summed by prelu / min-max / pool2_add / in_bwd / prelu_bwd / gate_prelu_bwd / pool2_bwd (tests/parity.py)."""
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='f32x3'))
    _assert(parity.fused_chain_checks(net))
    net.close()


def test_x3_ops_at_fp32_tolerances():
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='f32x3'))
    rows = []
    for case in parity.CONV_CASES:
        rows += parity.check_conv(net, case)
    for case in parity.DECONV_CASES:
        rows += parity.check_deconv(net, case)
    rows += parity.real_shape_checks(net) + parity.long_m_checks(net)
    # the classifier's / discriminators' production channel counts at small batch: ping-pong tiles with an idle second
    # group, split-K slabs, 768-channel rows (not a power of two), the limb filter-gradient kernel
    for nb in (2, 3):
        for case in [(f'u3 Conv_2 512->512 @16 nb{nb}', nb, 16, 16, 512, 512, 3, 1, 0, True),
                     (f'u3 Conv_1 256->512 @16 nb{nb}', nb, 16, 16, 256, 512, 3, 1, 0, True),
                     (f'u3 Conv_3 1x1 256->512 @16 nb{nb}', nb, 16, 16, 256, 512, 1, 1, 0, True),
                     (f'u4 Conv_2 768->768 @8 nb{nb}', nb, 8, 8, 768, 768, 3, 1, 0, True),
                     (f'u1 Conv_2 128->128 @64 nb{nb}', nb, 64, 64, 128, 128, 3, 1, 0, True),
                     (f'Dp conv_3 4x4s2 128->256 @32 nb{nb}', nb, 32, 32, 128, 256, 4, 2, 0, False)]:
            rows += parity.check_conv(net, case, seed=5)
    # the discriminators' 4x4 stride-2 forward layers in the space-to-depth halo form (k_gemm_x3h.hip, S2D): shapes large
    # enough for the cost model to pick the 256-pixel tiles -- un-split, split K with K chunks crossing the 4 phases,
    # 8 images of 4x8 outputs per tile
    for case in [('Dp conv_1 4x4s2 64->128 @64 nb64 (s2d halo)', 64, 64, 64, 64, 128, 4, 2, 0, False),
                 ('Dp conv_3 4x4s2 128->256 @32 nb64 (s2d halo, split K)', 64, 32, 32, 128, 256, 4, 2, 0, False),
                 ('D conv_4 4x4s2 256->512 8x16 nb64 (s2d halo, 8 images per tile)', 64, 8, 16, 256, 512, 4, 2, 0, False)]:
        rows += parity.check_conv(net, case, seed=7)
    # input gradients towards a 64-channel tensor through a 4x4 stride-2 layer: the phase-PAIR form of the halo kernel (k_gemm_x3h.hip:
    # PAIR) -- 16 x 16 tiles, 4 images of 8 x 8, 16 images of 4 x 4 per tile; with K split over the channel chunks
    for case in [('pair 4x4s2 64->128 @64 nb8', 8, 64, 64, 64, 128, 4, 2, 0, False),
                 ('pair 4x4s2 64->256 @16 nb8', 8, 16, 16, 64, 256, 4, 2, 0, False),
                 ('pair 4x4s2 64->128 @8 nb32', 32, 8, 8, 64, 128, 4, 2, 0, False),
                 ('pair 4x4s2 64->512 @32 nb4 (split K)', 4, 32, 32, 64, 512, 4, 2, 0, False)]:
        rows += parity.check_conv(net, case, seed=13)
    # the row-halo filter-gradient kernel (k_gemm_x3.hip: wgrad_x3r -- 3x3 stride 1, 4x4 / 5x5 stride 2, 64-channel units)
    # at its edges: 8 image rows per K step across 3 images, ragged output-channel tile, 192 channels (3 blocks), the
    # 32-pixel segments of a 128-wide row, reflect and zero padding, odd image counts
    for case in [('rowhalo 3x3 REFLECT C128->128 4x4 nb3', 3, 4, 4, 128, 128, 3, 1, 1, True),
                 ('rowhalo 3x3 C64->132 16x8 nb5', 5, 16, 8, 64, 132, 3, 1, 0, False),
                 ('rowhalo 3x3 REFLECT C192->128 8x16', 2, 8, 16, 192, 128, 3, 1, 1, True),
                 ('rowhalo 3x3 C64->256 128x128', 1, 128, 128, 64, 256, 3, 1, 0, False),
                 ('rowhalo 4x4s2 C64->128 8x8 nb3', 3, 8, 8, 64, 128, 4, 2, 0, False),
                 ('rowhalo 4x4s2 C128->132 16x8 nb5', 5, 16, 8, 128, 132, 4, 2, 0, False)]:
        rows += parity.check_conv(net, case, seed=11)
    rows += parity.check_deconv(net, ('rowhalo 5x5s2 256->128 4x8 nb3', 3, 4, 8, 256, 128), seed=11)
    net.close()
    assert len(rows) > 170
    _assert(rows)
    assert max(r[1] / r[2] for r in rows) < 0.2, 'f32x3 op errors are expected an order of magnitude inside the fp32 bounds'


@pytest.mark.parametrize('num_classes', [14, 0])
def test_x3_network_forwards(num_classes):
    test_step_gpu.test_network_forwards(num_classes)


@pytest.mark.parametrize('flags', [dict(), dict(E_norm='batch', G_norm='batch'), dict(E_norm=None, G_norm=None)],
                         ids=['default', 'batch_norms', 'no_norms'])
def test_x3_inference_matches_oracle(flags):
    test_step_gpu.test_inference_matches_oracle(flags)


def test_x3_substeps_14_classes():
    test_step_gpu.test_substeps_14_classes()


def test_x3_substeps_single_class_batch4():
    test_step_gpu.test_substeps_single_class_batch4()


@pytest.mark.parametrize('flags', [dict(), dict(E_norm='batch', G_norm='batch', D_norm='batch')], ids=['default', 'batch_norms'])
def test_x3_train_step_equals_sequential_substeps_and_is_deterministic(flags):
    test_step_gpu.test_train_step_equals_sequential_substeps_and_is_deterministic(flags)


@pytest.mark.parametrize('name', list(test_tight_gpu.SF.LOSSES))
def test_x3_substep_gradients_at_1e4_on_margin_selected_fixtures(name):
    test_tight_gpu.test_substep_gradients_at_1e4_on_margin_selected_fixtures(name)


@pytest.mark.parametrize('name', test_tight_gpu.FULL_BATCH_SUBSTEPS)
def test_x3_full_batch_substep_against_fp64_oracle(name):
    test_tight_gpu.test_full_batch_fp32_substep_against_fp64_oracle(name)


def test_x3_committed_goldens():
    test_golden_gpu.test_ops_match_committed_goldens()
    for fixture in test_golden_gpu.MG.NET_FIXTURES:
        test_golden_gpu.test_network_forwards_match_committed_goldens(fixture)
    test_golden_gpu.test_inference_on_example_sketches_matches_committed_goldens()


def test_x3_odd_batch_with_the_kink_flip_accounted_for():
    """The batch-3 fixture of test_step_gpu.py::test_substeps_odd_batch has, in the oracle itself, a classifier PReLU
    pre-activation (unit 3, Conv_1: 393216 elements) at 1.4e-7 of the layer's rms -- inside fp32 rounding of the kink.
    Whichever side an fp32 implementation lands on moves every gradient upstream of it by ~1e-3 in relative L2 (inside
    the 1e-2 tolerance) and, in absolute terms, by ~1e-5 of the sub-step's largest gradient: a scalar PReLU slope whose
    own gradient is only 1e-4 of that cannot be held to a RELATIVE bound there.  Every row must pass as in the fp32
    test, except that a scalar variable is judged by its absolute error against 3e-5 of the sub-step's largest gradient."""
    cfg = ea.Config(batch_size=3)
    values = ea.init_variables(cfg, 11)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=3), share_oracle=True)   # (test_substeps_odd_batch's fixture)
    net.close()
    fixed = []
    for name, err, tol in rows:
        if err > tol and 'prelu/param' in name and '|g|/gmax' in name:
            frac = float(name.split('|g|/gmax')[1].strip(' ]'))       # |g| / gmax of this scalar
            fixed.append((name + ' (absolute: err * |g|/gmax vs 3e-5)', err * max(frac, 1e-5), 3e-5))
        else:
            fixed.append((name, err, tol))
    _assert(fixed)


def test_x3_full_size_classifier_step_equals_mean_of_small_shards():
    """test_step_gpu.py's size-independent property (sub-step 4 at B=64 == mean over 32 shards of 2).  The shards run
    the 64x64 limb kernel, the full batch the ping-pong kernel with split K: different summation orders, hence
    independent kink decisions on the ~1e8 PReLU / min-max gate inputs of the batch.  The exact-fp32 kernels measure
    1e-6 (layers behind the last kink) .. 5e-4, f32x3 1e-6 .. 1.4e-3; the bound here is 3e-3, and layers downstream of
    every kink (unit 4, the classifier head) must agree to 1e-5."""
    names = ['D2/Conv/weights', 'D2/mru_conv_unit_t_1_layer_0/Conv_2/weights', 'D2/mru_conv_unit_t_2_layer_0/Conv_1/weights',
             'D2/mru_conv_unit_t_3_layer_0/Conv_3/biases', 'D2/mru_conv_unit_t_4_layer_0/Conv_2/weights',
             'D2/fully_connected/weights']
    big_cfg, small_cfg = ea.Config(batch_size=64), ea.Config(batch_size=2)
    values = ea.init_variables(big_cfg, 0)
    batch = ea.synthetic_batch(big_cfg)
    big = ea.EdgeGAN(big_cfg, values)
    big.upload_batch(**batch)
    big.substep(4, apply=False)
    want = {n: big.get_variable(n + '/grad').astype(np.float64) for n in names}
    big.close()
    small = ea.EdgeGAN(small_cfg, values)
    acc = {n: 0.0 for n in names}
    for s in range(32):
        sl = slice(2 * s, 2 * s + 2)
        small.upload_batch(batch['images'][sl], batch['z'][sl], batch['alpha'][:, sl].copy(), batch['eps'])
        small.substep(4, apply=False)
        for n in names:
            acc[n] = acc[n] + small.get_variable(n + '/grad').astype(np.float64) / 32
    small.close()
    for n in names:
        tol = 1e-5 if ('unit_t_4' in n or 'fully_connected' in n) else 3e-3
        assert parity.rel_l2(want[n], acc[n]) < tol, (n, parity.rel_l2(want[n], acc[n]))
