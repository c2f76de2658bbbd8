"""Full-size sub-step parity without minutes of oracle time per test: the fp64 oracle's gradients at BASELINE's full batch
(and the bf16-emulating oracle's at batch 16 / at configs[4]'s geometry) are committed as DIGESTS -- tests/golden/
full_batch.npz, written by tests/golden/make_full_batch_goldens.py from oracle/edgegan_oracle.py -- and the GPU tests
compare the engine's gradients with them:

  * a variable with <= EXACT_MAX elements is stored whole (fp64): its relative L2 error is exact;
  * a larger one is stored as K Rademacher projections <g, r_k> (r_k in {-1, +1}^n from a counter-based generator seeded by
    the variable's name) plus its L2 norm.  For any vector e, E_r <e, r>^2 = ||e||^2, so
        ||h - g||  ~=  sqrt(mean_k (<h, r_k> - <g, r_k>)^2)        (relative standard error 1 / sqrt(2K) = 12.5 % at K = 32)
    -- an unbiased estimate of the same relative-L2 figure the live comparison computes, from 33 numbers instead of
    up to 17 M.  An error concentrated in a few elements (a ReLU-kink flip) and a diffuse one (a wrong summation) are both
    seen, with their true L2 weight;
  * per variable the oracle's OWN fp32-vs-fp64 deviation (the tolerance rule of tests/test_tight_gpu.py), per sub-step the
    losses, the largest gradient element and whether the oracle's kink recorder found a small-extent activation inside MARGIN.

EG_LIVE_ORACLE=1 makes the tests evaluate the oracle themselves (exact relative L2 on every variable; minutes per sub-step).
tests/test_golden_cpu.py re-derives one entry from the oracle so that a silent edit of either side fails on CPU."""
import hashlib
import os

import numpy as np

K_PROJ = 32
EXACT_MAX = 4096
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'full_batch.npz')

# (key, sub-step, engine config kwargs, variable seed, batch seed, bf16 emulation)
CASES = [(f'b64/{n}', n, dict(batch_size=64), 0, 1234, False)
         for n in ('d_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim')] + \
        [('b64/g_optim_b', 'g_optim_b', dict(batch_size=64), 1, 1, False)] + \
        [(f'bf16_b16/{n}', n, dict(batch_size=16), 5, 21, True) for n in ('d_optim', 'd_optim2', 'g_optim_u', 'd_optim_patch3')] + \
        [(f'bf16_256x512_b2/{n}', n, dict(batch_size=2, height=256, width=512), 5, 21, True) for n in ('d_optim2', 'd_optim', 'g_optim_u')] + \
        [('bf16_b64/d_optim2', 'd_optim2', dict(batch_size=64), 5, 21, True)]      # configs[3]'s per-GPU batch
CASE = {c[0]: c for c in CASES}


def project(name, v):
    """<v, r_k>, k < K_PROJ, in fp64."""
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    out = np.zeros(K_PROJ)
    step = 1 << 18
    for lo in range(0, v.size, step):
        hi = min(v.size, lo + step)
        out += v[lo:hi] @ _signs_fast(name, lo, hi)
    return out


_RNG_CACHE = {}


def _signs_fast(name, lo, hi):
    """Rows lo..hi (inside one 2^18-row block) of the [n, K_PROJ] sign matrix of variable `name`: bits of numpy's Philox stream
    (counter-based, specified bit for bit), keyed by a hash of the name, counter = the block index."""
    assert lo % (1 << 18) == 0 and hi - lo <= (1 << 18)
    key = int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], 'little')
    g = np.random.Generator(np.random.Philox(key=key, counter=lo >> 18))
    bits = np.unpackbits(np.frombuffer(g.bytes((1 << 18) * K_PROJ // 8), dtype=np.uint8)).reshape(-1, K_PROJ)[: hi - lo]
    return bits.astype(np.float64) * 2.0 - 1.0


def digest(name, g):
    g = np.asarray(g, dtype=np.float64)
    if g.size <= EXACT_MAX:
        return {'exact': g.reshape(-1)}
    return {'proj': project(name, g), 'norm': np.array([np.linalg.norm(g.reshape(-1))]), 'absmax': np.array([np.abs(g).max()])}


def rel_l2_vs_digest(name, h, entry, floor):
    """relative L2 error of the engine's h against the digested oracle gradient; floor as parity.rel_l2's (per element)."""
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    if not np.all(np.isfinite(h)):
        return float('inf'), float('nan')
    if 'exact' in entry:
        g = entry['exact']
        if g.size != h.size:
            return float('inf'), float('nan')
        den = max(np.linalg.norm(g), floor * np.sqrt(g.size), 1e-300)
        return float(np.linalg.norm(h - g) / den), float(np.abs(g).max())
    d = project(name, h) - entry['proj']
    den = max(float(entry['norm'][0]), floor * np.sqrt(h.size), 1e-300)
    return float(np.sqrt(np.mean(d * d)) / den), float(entry['absmax'][0])


_LOADED = None


def load(case_key):
    """-> dict(vars={vn: entry}, self_dev={vn: float}, kappa={slope vn: float}, losses={k: float}, gmax=float, near_kink=bool)"""
    global _LOADED
    if _LOADED is None:
        _LOADED = np.load(GOLDEN)
    z, pre = _LOADED, case_key + '|'
    out = dict(vars={}, self_dev={}, kappa={}, losses={}, gmax=float(z[pre + 'gmax'][0]), near_kink=bool(z[pre + 'near_kink'][0]))
    for k in z.files:
        if not k.startswith(pre):
            continue
        parts = k[len(pre):].split('|')
        if parts[0] == 'v':
            out['vars'].setdefault(parts[1], {})[parts[2]] = z[k]
        elif parts[0] in ('self_dev', 'kappa'):
            out[parts[0]][parts[1]] = float(z[k][0])
        elif parts[0] == 'loss':
            out['losses'][parts[1]] = float(z[k][0])
    return out
