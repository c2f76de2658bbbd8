"""Tight sub-step parity: (1) margin-selected fixtures at relative L2 1e-4, (2) BASELINE's full batch (B=64, fp32)
against the fp64 oracle.

(1) tests/test_step_gpu.py asserts sub-step gradients at relative L2 1e-2 because ReLU / LeakyReLU / PReLU derivatives
jump at 0 and a fixture with ~1e6 activations always has one within fp32 rounding of the kink in some small-extent
layer (one flip there moves a channel's gradient by ~1/(B*H*W)).  That argument is testable: the fixtures used here
were picked (tests/golden/select_fixtures.py) so that every piecewise-linear activation with a normalisation extent
<= 256 keeps |pre-activation| >= MARGIN * rms -- re-derived below from the oracle, not trusted from the file -- and on
them every optimised variable of every sub-step must agree with fp64 autograd to 1e-4 in relative L2 (a systematic
0.5 % error would fail).
(2) The 128x128 tiles, split-K slabs, 84-way wgrad splits and flattened-(tap,c) kernels of the production shapes
(models/discriminator.py:61-72, models/generator.py:54-68, models/encoder.py:54-84, models/classifier.py) only occur at
B=64; sub-steps 1, 3, 4, 5 and 6 are compared there against the fp64 oracle on every optimised variable (sub-step 2 is
sub-step 3 on the other image half; sub-step 7 is sub-step 5 after the encoder's update)."""
import importlib.util
import json
import os
import re

import numpy as np
import pytest
import torch

import edgegan_amd as ea
import edgegan_oracle as O
import full_batch_lib as FB
import parity

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
spec = importlib.util.spec_from_file_location('select_fixtures', os.path.join(GOLDEN, 'select_fixtures.py'))
SF = importlib.util.module_from_spec(spec)
spec.loader.exec_module(SF)

MARGIN = 3e-6       # |pre-activation| / rms of the closest small-extent activation: ~30 fp32 ulps of the layer's scale
SUBSTEP_INDEX = {n: i for i, n in enumerate(O.SUBSTEPS, start=1)}


def _grad_rows(net, om, inp, name, tol, tag, var_tol=None, scalar_tol=None, precomputed=None):
    grads, losses = precomputed or om.compute_grads(name, inp, inp.eps[0])
    net.substep(SUBSTEP_INDEX[name], apply=False)
    gmax = max(float(g.abs().max()) for g in grads.values())
    rows = []
    for vn, g in grads.items():
        h = net.get_variable(vn + '/grad')
        if float(g.abs().max()) < 1e-9 * gmax:       # structural zero (bias in front of an InstanceNorm)
            rows.append((f'{tag} {name} {vn} (structural zero: noise / 1e-5 gmax)', float(np.abs(h).max()) / (1e-5 * gmax), 1.0))
        else:
            mx = parity.relerr(h, g.numpy(), 1e-5 * gmax)
            t = max(tol, (var_tol or {}).get(vn, 0.0))
            if scalar_tol and g.numel() == 1:
                t = max(t, scalar_tol)
            rows.append((f'{tag} {name} grad {vn} [L2; max-norm {mx:.1e}]', parity.rel_l2(h, g.numpy(), 1e-5 * gmax), t))
    hl = net.losses()
    for lk in SF.LOSSES[_LOSS_OF.get(name, name)]:
        ref = float(losses[lk])
        rows.append((f'{tag} {name} loss {lk}', abs(hl[lk] - ref) / abs(ref), 2e-5))
    return rows


_LOSS_OF = {'g_optim_b': 'g_optim_u'}      # (the same minimize ops, edgegan.py:117-124)


def _grad_rows_digest(net, name, fb, tol, tag, var_tol=None, loss_tol=2e-5):
    """_grad_rows against the committed digests of the oracle's gradients (tests/full_batch_lib.py): exact relative L2 for the
    small variables, the projection estimate of it for the large ones."""
    net.substep(SUBSTEP_INDEX[name], apply=False)
    gmax = fb['gmax']
    rows = []
    for vn, entry in fb['vars'].items():
        h = net.get_variable(vn + '/grad')
        err, absmax = FB.rel_l2_vs_digest(vn, h, entry, 1e-5 * gmax)
        if absmax < 1e-9 * gmax:       # structural zero (bias in front of an InstanceNorm)
            rows.append((f'{tag} {name} {vn} (structural zero: noise / 1e-5 gmax)', float(np.abs(h).max()) / (1e-5 * gmax), 1.0))
        else:
            t = max(tol, (var_tol or {}).get(vn, 0.0))
            if 'exact' in entry:
                kind = 'exact L2'
            else:       # the estimate's relative standard error is 1 / sqrt(2 K): two of them on top of the bound
                kind = f'L2 from {FB.K_PROJ} projections, bound x {1 + 2 / np.sqrt(2 * FB.K_PROJ):.2f}'
                t *= 1.0 + 2.0 / np.sqrt(2.0 * FB.K_PROJ)
            rows.append((f'{tag} {name} grad {vn} [{kind}]', err, t))
    hl = net.losses()
    for lk in SF.LOSSES[_LOSS_OF.get(name, name)]:
        ref = fb['losses'][lk]
        rows.append((f'{tag} {name} loss {lk}', abs(hl[lk] - ref) / abs(ref), loss_tol))
    return rows


def _assert(rows, report):
    if os.environ.get('EG_TIGHT_REPORT'):
        print(report, 'worst rows:\n' + parity.format_rows(sorted(rows, key=lambda r: -r[1] / r[2])[:8]))
    bad = [r for r in rows if not r[1] <= r[2]]
    assert not bad, '\n' + parity.format_rows(bad)


@pytest.mark.parametrize('name', list(SF.LOSSES))
def test_substep_gradients_at_1e4_on_margin_selected_fixtures(name):
    fx = json.load(open(os.path.join(GOLDEN, 'margin_fixtures.json')))[name]
    cfg = ea.Config(batch_size=2)
    margin = SF.fixture_margin(cfg, fx['variable_seed'], fx['batch_seed'], name)
    assert margin >= MARGIN, f'fixture for {name} has margin {margin:.2e} < {MARGIN:.0e}: re-run select_fixtures.py'
    values = ea.init_variables(cfg, fx['variable_seed'])
    b = ea.synthetic_batch(cfg, seed=fx['batch_seed'])
    net = ea.EdgeGAN(cfg, values)
    net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
    # PReLU slopes are scalars: their gradient is ONE sum of dy*x over every negative element of a 64x64 .. 8x8 map
    # with mixed signs (condition number sum|t| / |sum t| of a few hundred), fp32 accumulation is worth ~1e-4 there
    rows = _grad_rows(net, om, inp, name, 1e-4, 'margin', scalar_tol=5e-4)
    net.close()
    assert len(rows) >= 7
    _assert(rows, f'{name} (margin {margin:.1e})')


# Tolerance of the full-batch rows: max(1e-3, SELF_DEV_FACTOR x the oracle's own fp32-vs-fp64 deviation on this fixture).  The
# general factor is 2 (it had been raised to 3 right after two rows measured 2.04 x and 2.1 x).  The rows that need
# more are listed by name with what was measured (profiles/r04n_tight_ratios.txt: every variable's error / self-deviation in
# both fp32 modes; everything else is <= 1.9 x or below the 1e-3 floor): the two image-side thin convolutions of the classifier's
# stem -- 3 -> 8 channels over 64 x 64 maps, each filter element a sum over 1.7e7 pixel terms of mixed sign behind ~1e8 PReLU /
# min-max kinks -- where the oracle's fp32 run (torch CPU, its summation order depends on the host's thread count) is ONE sample
# of the fp32 noise and the engine's another.
SELF_DEV_FACTOR = 2.0
# scalar PReLU slopes: tolerance kappa x this (kappa = sum|t_i| / |sum t_i| of the slope's gradient terms, _full_batch_oracle).
# Round 3 used 1e-4; the 28 slope rows of both modes measure <= 0.029 of that (r04n), so it is 1e-5 now: a >= 3.4 x margin.
SLOPE_KAPPA_TOL = 1e-5
SELF_DEV_OUTLIERS = {
    'D2/mru_conv_unit_t_1_layer_0/Conv/weights': 2.6,     # measured 2.04 (r03, two boxes), 2.12 (r04n)
    'D2/Conv/weights': 2.6,                               # measured 1.95 (r04n): inside 2 x on that box, not by a margin
}
# Kink flips in the full-batch rows.  At B = 64 the encoder's deep blocks (512 channels on 8x8 ... 2x2 maps: InstanceNorm over
# 64 ... 4 pixels) hold ~1e6 ReLU inputs per sub-step and the fp64 oracle itself finds some within fp32 rounding of the kink
# (e_optim on this fixture: |x| / rms = 2.5e-7, 1.1e-6, 1.6e-6, 2.4e-6 -- MARGIN is 3e-6).  Which side an fp32 evaluation lands
# on depends on its rounding, not on its correctness: the oracle's own fp32 run flips some (its deviation from fp64 on the two
# filters next to such an element: 1.6e-3 ... 2.8e-3), the engine's modes others (round 4: re-ordering the sums of the
# generators' 3-channel output layer moved E/e_resnet_512_4/res1 and res2 from 1.8e-3 / 7e-4 to 5.5e-3 / 5.0e-3 in the f32x3
# mode, error concentrated in a few elements: max-norm 10 x the L2 figure).  A flip costs the adjacent filters' gradients up to
# ~5e-3 in relative L2, a wiring error >= 1e-1.  Rounds 3 / 4 judged the residual-branch filters of the encoder's two deepest blocks
# at the ordinary sub-step tests' flip-tolerant 1e-2 whenever the oracle's fp64 pass recorded a small-extent ReLU input inside MARGIN.
# Round 5 replaced that tolerance by a proof in the runs that evaluate the oracle live: _flip_proof below re-judges such a
# row at the TIGHT tolerance against the oracle with the responsible near-kink elements mirrored (e_optim: two inputs of block
# 512_4's last ReLU at 2.4e-6 / 2.8e-6 of the layer's rms, 5.5e-3 -> 1.1e-5; d_optim_patch2: two LeakyReLU inputs of the last
# InstanceNorm layer, 1.2e-3 -> 1.3e-5).  Round 6: BOTH fp32 modes evaluate the oracle live (~10 s per
# sub-step since torch is capped at the container's CPU quota), so the flip tolerances FLIP_TOL = 1e-2 / FLIP_TOL_D = 3e-3 of the
# digest-based runs are gone: a flip-exposed row either meets the tight tolerance or is explained, element by element, by the proof.
FLIP_EXPOSED = re.compile(r'^E/e_resnet_(256_2|512_[34])/res[12]/conv2d/w$')   # InstanceNorm extents 256 / 64 / 16 pixels (<= SF.SMALL_EXTENT)
# Round 5 added d_optim_patch2 (so far argued from d_optim_patch3): the discriminators' InstanceNorm + LeakyReLU layers behind
# d_conv_3 / d_conv_4 have extents of 256 / 64 pixels (joint D: 128 / 32) -- a flip there moves every filter at and upstream of
# it.  Measured: on d_optim_patch3 the ORACLE'S OWN fp32 run deviates from fp64 by 1.1e-3 / 2.1e-3 / 2.4e-3 on d_conv_0 / 1 / 3
# (its flips), on d_optim_patch2 by <= 1e-4 (none on the generating host) while the f32x3 engine measures 1.2e-3 / 1.3e-3 /
# 1.2e-3 on d_conv_1 / 3 / 4 with the d_conv_4 error concentrated in a few elements (max-norm 14 x the L2 figure: a flip's
# signature).  Same rule: a discriminator filter that misses the tight tolerance goes through the proof.
FLIP_EXPOSED_D = re.compile(r'^D(_patch[23])?/d_conv_[0134]/conv2d/w$')
# All seven sub-steps at B = 64 (round 5: d_optim_patch2 and g_optim_b were argued by symmetry before).  g_optim_b is the same
# graph as g_optim_u (edgegan.py:117-124), so it is evaluated on a DIFFERENT state -- variable seed 1, batch seed 1 -- to be an
# independent comparison of the engine's sub-step 7.  The exact-fp32 mode (this file) runs the two sub-steps with the most
# kernels of its own -- the joint discriminator (gradient penalty) and the classifier; the benched mode runs all seven
# (tests/test_x3_gpu.py), sharing the oracle's work where both run.
FULL_BATCH_SUBSTEPS = ['d_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim', 'g_optim_b']
FULL_BATCH_EXACT_FP32 = ['d_optim', 'd_optim2']
_FULL_BATCH_ORACLE = {}     # name -> (g64, l64, self_dev): the f32x3 re-run of these tests (test_x3_gpu.py) reuses the oracle's work


def _full_batch_oracle(name, cfg, values, b, key=None):
    key = key or name
    if key not in _FULL_BATCH_ORACLE:
        om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
        inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
        om32 = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float32)
        inp32 = O.StepInputs(*[torch.tensor(np.asarray(b[k]), dtype=torch.float32) for k in ('images', 'z', 'alpha', 'eps')])
        g32, _ = om32.compute_grads(name, inp32, inp32.eps[0])
        O._SLOPE_TERMS = {}          # record sum_i |t_i| of every PReLU slope's gradient terms (conditioning, below)
        try:
            with O.kink_recorder(collect=(SF.SMALL_EXTENT, FLIP_CANDIDATE_MARGIN)) as rec:
                g64, l64 = om.compute_grads(name, inp, inp.eps[0])
        finally:
            O._SLOPE_TERMS = None
        near_kink = rec.margin(SF.SMALL_EXTENT) < MARGIN
        gmax = max(float(g.abs().max()) for g in g64.values())
        self_dev = {vn: parity.rel_l2(g32[vn].double().numpy(), g.numpy(), 1e-5 * gmax) for vn, g in g64.items()}
        # A scalar PReLU slope's gradient is ONE cancelling sum  sum_i t_i  (t_i = dy_i x_i over the layer's non-positive
        # inputs: 1e5 .. 1.7e7 terms of both signs).  Element-wise relative noise eps in the terms moves the SUM by up to
        # eps * kappa relative to itself, kappa = sum|t_i| / |sum t_i| (measured in the fp64 oracle: 12 .. 8e3 for the 14
        # slopes).  Its tolerance is therefore kappa x SLOPE_KAPPA_TOL (1e-5: a hundredth of the 1e-3 the tensors are held to)
        # where that exceeds the general rule; expressed through the same per-variable table.
        for vn, l1 in getattr(om, 'last_slope_l1', {}).items():
            if vn in g64:
                kappa = l1 / max(abs(float(g64[vn])), 1e-300)
                self_dev[vn] = max(self_dev[vn], SLOPE_KAPPA_TOL * kappa / SELF_DEV_FACTOR)
        _FULL_BATCH_ORACLE[key] = ({k: v.detach() for k, v in g64.items()}, {k: float(v) for k, v in l64.items()}, self_dev, near_kink)
        _FULL_BATCH_NEAR[key] = sorted(rec.near, key=lambda r: r[2])
    return _FULL_BATCH_ORACLE[key]


# The flip tolerance, proven where it is used: when a flip-exposed variable misses its TIGHT tolerance, the deviation
# must be the signature of single mask decisions.  Measured on d_optim_patch2 (tools/gpu_flip_probe.py, profiles/r05v_flip_probe.txt):
# 98 % of the error of d_conv_4's filter gradient sits in ONE of its 512 output channels, and flipping ONE input of that channel's
# LeakyReLU (|x| / rms = 5.0e-6 in the fp64 oracle; the layer's InstanceNorm runs over 16 pixels) in the oracle moves the error
# from 1.17e-3 to 2.3e-4 -- while the eight globally closest-to-kink elements explain nothing.  So the candidates are chosen where
# the error is: the oracle's small-extent activation inputs inside FLIP_CANDIDATE_MARGIN of the kink (relative to the layer's rms)
# that lie in the worst output channels of a failing filter (same channel count), at most FLIP_MAX_CANDIDATES of them; the oracle
# is re-evaluated with each mirrored to the other side of the kink (oracle/edgegan_oracle.py: kink_flipper -- a value change of
# ~1e-6 of the layer's scale; the masks, also those of dependent activations downstream, follow from the value), and the engine must meet the tight tolerance against  oracle + sum of the flip differences  for SOME subset, the same
# subset for every failing variable of the sub-step (the differences add: each flip changes one mask element of a
# piecewise-linear graph).  A deviation that no such subset explains is not a kink flip, whatever its size, and fails.
FLIP_CANDIDATE_MARGIN = 3e-5
FLIP_MAX_CANDIDATES = 6
_FULL_BATCH_NEAR = {}


def _flip_proof(name, cfg, values, b, g64, need, tight, gmax, key=None):
    """need: {failing flip-exposed variable: its gradient from the engine}; tight: {variable: tolerance}.  Returns (flipped
    candidates, {variable: error against the flipped oracle}, all candidates)."""
    import itertools
    chans = {}      # channel count -> worst output channels of the failing filters with that many
    for vn, h in need.items():
        g = g64[vn].numpy()
        if g.ndim != 4:
            continue
        per_n = np.sqrt(((h - g) ** 2).reshape(-1, g.shape[-1]).sum(0))
        chans.setdefault(g.shape[-1], set()).update(int(n) for n in np.argsort(-per_n)[:2])
    cand = [r for r in _FULL_BATCH_NEAR.get(key or name, []) if len(r[4]) == 4 and r[4][-1] in chans and int(r[1] % r[4][-1]) in chans[r[4][-1]]]
    cand = cand[:FLIP_MAX_CANDIDATES]
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float64)
    inp = O.StepInputs(parity.t64(b['images']), parity.t64(b['z']), parity.t64(b['alpha']), parity.t64(b['eps']))
    deltas = []
    for call, idx, _, _, _ in cand:
        with O.kink_flipper([(call, idx)]):
            gf, _ = om.compute_grads(name, inp, inp.eps[0])
        deltas.append({vn: (gf[vn] - g64[vn]).numpy() for vn in need})
    best = None
    for r in range(len(cand) + 1):
        for sub in itertools.combinations(range(len(cand)), r):
            errs = {}
            for vn, h in need.items():
                ref = g64[vn].numpy() + sum((deltas[i][vn] for i in sub), np.zeros_like(g64[vn].numpy()))
                errs[vn] = parity.rel_l2(h, ref, 1e-5 * gmax)
            score = max(errs[vn] / tight[vn] for vn in need)
            if best is None or score < best[0]:
                best = (score, sub, errs)
    return [cand[i] for i in best[1]], best[2], cand


def _rows_with_flip_proof(net, name, cfg, values, b, g64, l64, var_tol, tight_tol, tag, exposed, key=None):
    """Every optimised variable of sub-step `name` against the fp64 gradients g64 at its tight tolerance; a flip-exposed variable
    (exposed(vn)) that misses it must be explained by mirrored near-kink inputs of the oracle (_flip_proof): its row is then REPLACED
    by one at the same tolerance against the oracle with those elements flipped -- or stays failing."""
    rows = _grad_rows(net, None, None, name, 1e-3, tag, var_tol=var_tol, precomputed=(g64, l64))
    gmax = max(float(g.abs().max()) for g in g64.values())
    need = {}
    for vn in var_tol:
        if exposed(vn) and vn in g64 and float(g64[vn].abs().max()) >= 1e-9 * gmax:
            h = net.get_variable(vn + '/grad').astype(np.float64)
            if parity.rel_l2(h, g64[vn].numpy(), 1e-5 * gmax) > tight_tol[vn]:
                need[vn] = h
    if need:
        flipped, errs, cand = _flip_proof(name, cfg, values, b, g64, need, tight_tol, gmax, key=key)
        what = ', '.join(f'{c[3]} {c[4]} element {c[1]} at {c[2]:.1e} rms' for c in flipped) or 'none'
        if os.environ.get('EG_FLIP_REPORT'):
            print(tag, name, 'flip proof:', len(cand), 'candidates; flipped:', what,
                  {vn: ('%.1e' % parity.rel_l2(need[vn], g64[vn].numpy(), 1e-5 * gmax), '%.1e' % errs[vn], '%.1e' % tight_tol[vn]) for vn in need})
        # the proof row REPLACES the failing row: tight tolerance, against the oracle with the listed elements mirrored
        rows = [r for r in rows if not any(f' grad {vn} [' in r[0] for vn in need)]
        for vn in need:
            rows.append((f'{tag} {name} grad {vn} against the oracle with {len(flipped)} of {len(cand)} near-kink candidates flipped ({what}) [L2]',
                         errs[vn], tight_tol[vn]))
    return rows


@pytest.mark.parametrize('name', FULL_BATCH_EXACT_FP32)
def test_full_batch_fp32_substep_against_fp64_oracle(name):
    """B=64, 14 classes, fp32 (BASELINE configs[2]) -- every optimised variable of the sub-step against fp64 autograd, incl.
    the classifier sub-step d_optim2 (edgegan/models/classifier.py:50-115: 54 % of the iteration's FLOPs; every one of its
    58 variables, among them Conv_2/weights of all four units and the update_gate filters).  Tolerance per variable:
    max(1e-3, 2 x the ORACLE'S OWN fp32-vs-fp64 deviation on this very fixture) in relative L2 (SELF_DEV_FACTOR; two named
    outliers at 2.6 x, SELF_DEV_OUTLIERS) -- the rule of parity.substep_checks(self_consistency=True).  Filter gradients here are sums over 64 x H x W pixels (up to 1e6
    terms, mixed signs) and the batch holds ~30x more near-kink activations than the B=2 fixtures, so ANY fp32
    evaluation of this graph -- the reference's own included -- deviates from exact arithmetic by 1e-4 .. 3e-3; the
    oracle's fp32 run is ONE sample of that noise, the engine's another (measured on MI355X, exact-fp32 mode: d_optim
    <= 1.4e-4, d_optim_patch3 <= 3.6e-4, d_optim2 median 3e-4 with the two worst rows -- the unit-1 3->8-channel image
    convolution and a scalar PReLU slope, each a sum over 1.7e7 pixel terms behind 1e8 PReLU / min-max kinks -- at
    2.04 x and 2.1 x the oracle's own deviation of 5.9e-4 / 1.3e-3); a wiring error shows up as >= 1e-1.  Scalar PReLU
    slopes: tolerance scaled by the conditioning of their sum, see _full_batch_oracle (measured 1e-3 .. 5e-3)."""
    case = FB.CASE['b64/' + name]
    cfg = ea.Config(**case[2])
    values = ea.init_variables(cfg, case[3])
    b = ea.synthetic_batch(cfg, seed=case[4])
    # The benched mode (f32x3, re-run from tests/test_x3_gpu.py) evaluates the oracle HERE AND NOW -- exact relative L2 on every
    # variable, ~10 s of CPU per sub-step; the exact-fp32 mode compares with the committed digests of that same evaluation
    # (tests/full_batch_lib.py).  EG_LIVE_ORACLE=1 / 0 forces one or the other.  Tolerances come from the digests' four fp32
    # samples in both cases (plus the live run's own sample): one fp32 run is one draw of the kink decisions.
    live = os.environ.get('EG_LIVE_ORACLE', '1') != '0'     # (0: against the committed digests, every row at the tight tolerance)
    fb = FB.load('b64/' + name)
    self_dev, near_kink = dict(fb['self_dev']), fb['near_kink']
    for vn, kappa in fb.get('kappa', {}).items():
        self_dev[vn] = max(self_dev[vn], SLOPE_KAPPA_TOL * kappa / SELF_DEV_FACTOR)
    if live:
        g64, l64, live_dev, live_kink = _full_batch_oracle(name, cfg, values, b)
        _FULL_BATCH_ORACLE.pop(name, None)      # (tens of MB of fp64 gradients each)
        self_dev = {vn: max(d, live_dev.get(vn, 0.0)) for vn, d in self_dev.items()}
        near_kink = near_kink or live_kink
    net = ea.EdgeGAN(cfg, values)
    net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
    var_tol = {vn: SELF_DEV_OUTLIERS.get(vn, SELF_DEV_FACTOR) * d for vn, d in self_dev.items()}
    tight_tol = {vn: max(1e-3, t) for vn, t in var_tol.items()}
    if live:
        rows = _rows_with_flip_proof(net, name, cfg, values, b, g64, l64, var_tol, tight_tol, 'B=64',
                                     lambda vn: bool(FLIP_EXPOSED.match(vn) or FLIP_EXPOSED_D.match(vn)))
    else:
        rows = _grad_rows_digest(net, name, fb, 1e-3, 'B=64', var_tol=var_tol)
    net.close()
    if os.environ.get('EG_TIGHT_RATIOS'):     # tools: every variable's error over the oracle's own fp32-vs-fp64 deviation
        with open(os.environ['EG_TIGHT_RATIOS'], 'a') as f:
            for r in rows:
                vn = r[0].split(' grad ')[-1].split(' [')[0] if ' grad ' in r[0] else None
                if vn in self_dev:
                    f.write(json.dumps({'mode': os.environ.get('EG_F32_MODE', 'f32'), 'substep': name, 'variable': vn, 'err': r[1],
                                        'self_dev': self_dev[vn], 'ratio': r[1] / max(self_dev[vn], 1e-30), 'tol': r[2]}) + '\n')
    if name == 'd_optim2':
        assert len(rows) >= 55 and sum('Conv_2/weights' in r[0] for r in rows) == 4 and sum('update_gate/weights' in r[0] for r in rows) == 4
    if os.environ.get('EG_TIGHT_REPORT'):
        print(name, 'oracle fp32-vs-fp64 self-deviation: median %.1e max %.1e' % (np.median(list(self_dev.values())), max(self_dev.values())))
    _assert(rows, name)
