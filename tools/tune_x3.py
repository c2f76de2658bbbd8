#!/usr/bin/env python
"""Per-shape choice of (tile family, K split) for the f32x3 convolution launches, and of the block target of the limb
filter-gradient kernels, from MEASUREMENTS instead of the cost model.

On the GPU box, with the experiments build (make -C edgegan_amd/csrc EXPERIMENTS=1):
    python tools/tune_x3.py sweep gpurun_out/<tag>        # one tools/layer_table.py run per candidate -> <tag>/tune_*.csv
Anywhere:
    python tools/tune_x3.py table gpurun_out/<tag> [min_gain]   # -> edgegan_amd/csrc/x3_tuned.h rows + a report on stdout

Every GEMM-shaped launch of an iteration is timed with HIP events (sub-steps serialised) under EG_X3_FORCE=<tile>,<ks>, which
puts EVERY limb-mode convolution on that candidate (clipped to what its shape allows); a launch shape is the tuple
(M, N, C, taps, nphase) -- for this model's layers that determines the stride (one phase with 16 or 25 taps is a stride-2
gather, everything else stride 1).  A shape gets a table row only where the best candidate beats the cost model's own choice
by min_gain (default 4 %) -- everything else keeps the model."""
import csv
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDS = [(1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (2, 1), (2, 2), (2, 4)]
WG_BLOCKS = [128, 192, 256, 384, 512]


def sweep(out):
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so')
    base = dict(os.environ, EG_LIB_PATH=lib, EG_X3_NO_TABLE='1')
    runs = [('default', {})] + [(f'c{t}_{k}', {'EG_X3_FORCE': f'{t},{k}'}) for t, k in CANDS] + \
           [(f'wg{b}', {'EG_WGRAD_X3_BLOCKS': str(b)}) for b in WG_BLOCKS if b != 256]
    for name, env in runs:
        csvp = os.path.join(out, f'tune_{name}.csv')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'layer_table.py'), csvp, '--steps', '2'],
                           env=dict(base, **env), capture_output=True, text=True)
        print(name, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else '', flush=True)


def load(path):
    rows = defaultdict(lambda: [0.0, 0.0, set()])     # shape -> [ms per iteration, launches per iteration, kernels]
    for r in csv.DictReader(open(path)):
        wg = r['layer'].endswith('wgrad')
        key = (wg, int(r['M']), int(r['N']), int(r['C']), int(r['taps']), int(r['nphase']))
        a = rows[key]
        a[0] += float(r['ms_per_iter'])
        a[1] += float(r['launches_per_iter'])
        a[2].add((r['kernel'], int(r['split'])))
    return rows


def table(out, min_gain=0.04):
    default = load(os.path.join(out, 'tune_default.csv'))
    cands = {(t, k): load(os.path.join(out, f'tune_c{t}_{k}.csv')) for t, k in CANDS
             if os.path.exists(os.path.join(out, f'tune_c{t}_{k}.csv'))}
    rows, saved, total = [], 0.0, 0.0
    for key, (ms, n, kern) in sorted(default.items(), key=lambda kv: -kv[1][0]):
        wg, M, N, C, taps, nph = key
        if wg or not any(k.startswith('conv_gemm_x3') for k, _ in kern):      # (only the launches pick_x3 decides)
            continue
        total += ms
        best, bc = ms, None
        for c, tab in cands.items():
            if key in tab and tab[key][0] < best:
                best, bc = tab[key][0], c
        kn = '/'.join(sorted(f'{k}:{s}' for k, s in kern))
        if bc and best < ms * (1.0 - min_gain) and cands[bc][key][2] != kern:
            got = '/'.join(sorted(f'{k}:{s}' for k, s in cands[bc][key][2]))
            ks = max(s for _, s in cands[bc][key][2])       # the split that actually ran (the force is clipped per shape)
            S = 2 if (nph == 1 and taps in (16, 25)) else 1
            rows.append((M, N, C, taps, nph, S, bc[0], ks))
            saved += ms - best
            print(f'# M={M:7d} N={N:4d} C={C:4d} taps={taps:2d} ph={nph} n={n:4.1f}  {ms:.3f} -> {best:.3f} ms  {kn}  ->  {got}')
    print(f'# limb convolutions {total:.2f} ms per iteration (serialised), table saves {saved:.2f} ms')
    hdr = '// (tile family 1 = 256-pixel halo / ping-pong kernels, 2 = 64 x 64): {M, N, C, taps, nphase, S, tile, ks},\n'
    with open(os.path.join(ROOT, 'edgegan_amd', 'csrc', 'x3_tuned.h'), 'w') as f:
        f.write(hdr)
        for r in rows:
            f.write('    {%d, %d, %d, %d, %d, %d, %d, %d},\n' % r)
    # filter gradients: one block target for all (the kernels' split is a balanced division of pixels, not per shape)
    print('# limb filter gradients, ms per iteration by block target:')
    for b in WG_BLOCKS:
        p = os.path.join(out, 'tune_default.csv' if b == 256 else f'tune_wg{b}.csv')
        if os.path.exists(p):
            t = load(p)
            print(f'#   {b:4d}: {sum(v[0] for k, v in t.items() if k[0]):.3f}')


def sweep_bf16(out):
    """The bf16 mode's plane-fed convolutions: K split 1 .. 4 forced on every launch (EG_BF16P_FORCE_KS), per-layer timings."""
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so')
    for name, env in [('bfdefault', {})] + [(f'bfks{k}', {'EG_BF16P_FORCE_KS': str(k)}) for k in (1, 2, 3, 4)]:
        csvp = os.path.join(out, f'tune_{name}.csv')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'layer_table.py'), csvp, '--dtype', 'bf16', '--steps', '2'],
                           env=dict(os.environ, EG_LIB_PATH=lib, **env), capture_output=True, text=True)
        print(name, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else '', flush=True)


def table_bf16(out, min_gain=0.05):
    default = load(os.path.join(out, 'tune_bfdefault.csv'))
    cands = {k: load(os.path.join(out, f'tune_bfks{k}.csv')) for k in (1, 2, 3, 4) if os.path.exists(os.path.join(out, f'tune_bfks{k}.csv'))}
    rows, saved, total = [], 0.0, 0.0
    for key, (ms, n, kern) in sorted(default.items(), key=lambda kv: -kv[1][0]):
        wg, M, N, C, taps, nph = key
        if wg or not any(k.startswith('conv_bf16p') for k, _ in kern):
            continue
        total += ms
        best, bk = ms, None
        for k, tab in cands.items():
            if key in tab and tab[key][0] < best:
                best, bk = tab[key][0], k
        if bk and best < ms * (1.0 - min_gain) and cands[bk][key][2] != kern:
            ks = max(s for _, s in cands[bk][key][2])
            S = 2 if (nph == 1 and taps in (16, 25)) else 1
            rows.append((M, N, C, taps, nph, S, 0, ks))
            saved += ms - best
            print(f'# M={M:7d} N={N:4d} C={C:4d} taps={taps:2d} ph={nph} n={n:4.1f}  {ms:.3f} -> {best:.3f} ms  '
                  f"{'/'.join(sorted(f'{k}:{s}' for k, s in kern))}  ->  ks {ks}")
    print(f'# plane-fed convolutions {total:.2f} ms per iteration (serialised), table saves {saved:.2f} ms')
    with open(os.path.join(ROOT, 'edgegan_amd', 'csrc', 'bf16p_tuned.h'), 'w') as f:
        f.write('// plane-fed bf16 convolutions: {M, N, C, taps, nphase, S, 0, ks},\n')
        for r in rows:
            f.write('    {%d, %d, %d, %d, %d, %d, %d, %d},\n' % r)


if __name__ == '__main__':
    cmd = sys.argv[1]
    if cmd == 'sweep':
        sweep(sys.argv[2])
    elif cmd == 'sweep_bf16':
        sweep_bf16(sys.argv[2])
    elif cmd == 'table_bf16':
        table_bf16(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.05)
    else:
        table(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.04)
