"""The oracle reproduces the committed golden vectors (tests/golden/*.npz, written by tests/golden/make_goldens.py).

Detects any silent edit of oracle/edgegan_oracle.py that changes a number.  Tolerance 1e-11 relative to the
largest element of each entry: bit-exact on the generating host, the slack only covers a different summation
order inside torch's threaded fp64 kernels on another host (core count dependent)."""
import importlib.util
import os

import numpy as np
import pytest

import golden_lib as G

spec = importlib.util.spec_from_file_location('make_goldens', os.path.join(G.GOLDEN_DIR, 'make_goldens.py'))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)

FILES = ['ops.npz'] + [f[0] for f in MG.NET_FIXTURES] + [f[0] for f in MG.STEP_FIXTURES] + [MG.INFER_FIXTURE[0]]
TOL = 1e-11


@pytest.mark.parametrize('fn', FILES)
def test_oracle_reproduces_goldens(fn):
    want = np.load(os.path.join(G.GOLDEN_DIR, fn))
    got = MG.build(fn)[fn]
    assert sorted(got.keys()) == sorted(want.files), 'golden key set changed: regenerate deliberately, with a reason'
    bad = []
    for k in want.files:
        w, g = want[k], np.asarray(got[k], dtype=np.float64)
        if w.shape != g.shape:
            bad.append((k, 'shape', w.shape, g.shape))
            continue
        scale = max(float(np.max(np.abs(w))) if w.size else 0.0, 1e-300)
        err = float(np.max(np.abs(w - g))) / scale if w.size else 0.0
        if not err <= TOL:
            bad.append((k, err))
    assert not bad, bad[:20]


def test_goldens_are_small_and_complete():
    total = sum(os.path.getsize(os.path.join(G.GOLDEN_DIR, f)) for f in FILES)
    assert total < 2 << 20
    st = np.load(os.path.join(G.GOLDEN_DIR, 'steps_14class_b2.npz'))
    # 2 iterations x 7 sub-steps, every optimised variable of each sub-step has a gradient digest
    for it in (0, 1):
        for name in ('d_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim', 'g_optim_b'):
            assert any(k.startswith(f'it{it}/{name}/grad/') for k in st.files), (it, name)
            assert any(k.startswith(f'it{it}/{name}/loss/') for k in st.files), (it, name)
    # the dead classifier head (quirk N8) never gets a gradient and never moves
    assert not any('D2/Conv_1/' in k and '/grad/' in k for k in st.files)
    assert st['it1/delta/D2/Conv_1/weights'][1] == 0.0


def test_full_batch_digest_is_rederived_from_the_oracle():
    """tests/golden/full_batch.npz (digests of the oracle's full-size sub-step gradients, tests/full_batch_lib.py) is what the
    GPU tests at B = 64 / bf16 B = 16 compare the engine with: one case -- the encoder's sub-step at BASELINE's full batch -- is
    re-evaluated here from oracle/edgegan_oracle.py and must reproduce its committed digest (exactly stored small variables,
    the Rademacher projections and norms of the large ones, the loss), so that neither side can drift silently.  The oracle's
    fp32 self-deviation entries depend on the host's thread count and are not compared."""
    import full_batch_lib as FB
    spec2 = importlib.util.spec_from_file_location('make_full_batch_goldens', os.path.join(G.GOLDEN_DIR, 'make_full_batch_goldens.py'))
    MF = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(MF)
    key = 'b64/e_optim'
    got = MF.entries(FB.CASE[key])
    want = np.load(FB.GOLDEN)
    checked = 0
    for k, g in got.items():
        parts = k.split('|')
        if parts[1] in ('self_dev', 'kappa', 'near_kink'):
            continue
        w = want[k]
        if parts[1] == 'v' and parts[3] == 'proj':
            scale = float(want['|'.join(parts[:3] + ['norm'])][0])      # a projection is a cancelling sum: compare against ||g||
        else:
            scale = max(float(np.max(np.abs(w))), 1e-300)
        noise = 1e-13 * float(want[key + '|gmax'][0])      # (structural zeros -- a bias in front of an InstanceNorm -- hold 1e-16 noise)
        assert w.shape == g.shape and float(np.max(np.abs(w - g))) <= 1e-9 * scale + noise, k
        checked += 1
    assert checked >= 60
    assert sorted(k for k in want.files if k.startswith(key + '|')) == sorted(got)
