#!/usr/bin/env python
"""Per-layer roofline table of one training iteration (BASELINE configs[2]: 14-class 64x128 fp32, batch 64).

Every GEMM-shaped launch of an iteration is bracketed by HIP events on the engine's stream (eg_profile_begin /
eg_profile_layers; sub-steps serialised so that kernels do not overlap) and labelled with the TF variable whose
filter it contracts with and the pass (fwd / dgrad / wgrad).  Output: one CSV row per (layer, pass, shape):
launches per iteration, average microseconds, algorithmic TFLOP/s, executed MFMA TFLOP/s (limb kernels: 6 bf16 MFMA flops per
fp32 flop) as a fraction of the MFMA peak of the instruction used (fp32 157.3 / bf16 2500 TFLOP/s) and, for the
HBM-class thin layers, algorithmic GB/s (input + output + filter bytes, fp32) against 8 TB/s.
usage: python tools/layer_table.py [out.csv] [--dtype bf16] [--steps 3]"""
import csv
import io
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import edgegan_amd as ea      # noqa: E402

PEAK_F32, PEAK_BF16, PEAK_HBM = 157.3, 2500.0, 8000.0


def main():
    out = next((a for a in sys.argv[1:] if a.endswith('.csv')), None)
    dtype = sys.argv[sys.argv.index('--dtype') + 1] if '--dtype' in sys.argv else 'f32x3'
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 3
    cfg = ea.Config(batch_size=64, compute_dtype=dtype)
    net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
    net.upload_batch(**ea.synthetic_batch(cfg))
    for _ in range(2):
        net.train_step()
    net.profile_begin()
    for _ in range(steps):
        net.train_step()
    text = net.profile_layers()
    net.profile_end()
    net.close()
    agg = OrderedDict()
    for r in csv.DictReader(io.StringIO(text)):
        key = (r['label'], r['kernel'], r['M'], r['N'], r['C'], r['taps'], r['nphase'], r['split'])
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r['us'])
        a[2] += float(r['gflop'])
    rows = []
    for (label, kern, M, N, C, taps, nph, split), (n, us, gf) in agg.items():
        M_, N_, C_, taps_ = int(M), int(N), int(C), int(taps)
        avg_us = us / n
        tf = gf / n / avg_us * 1e-3 if avg_us else 0.0         # GFLOP / us = PFLOP/s -> TFLOP/s
        tf = (gf / n) / (avg_us * 1e-6) / 1e3
        limb = 'x3' in kern       # limb kernels execute 6 bf16 MFMA flops per algorithmic fp32 flop (k_gemm_x3.hip)
        peak = PEAK_BF16 if ('bf16' in kern or limb) else PEAK_F32
        # algorithmic bytes: gathered operand + dense output + filter (fp32); for the multi-phase launches M is the
        # largest phase, the output grid is nphase * M
        nph_ = int(nph)
        byts = 4.0 * (M_ * nph_ * C_ / (4 if nph_ == 4 and 'dgrad' not in label and 'fwd' not in label else 1) + M_ * nph_ * N_ + taps_ * C_ * N_)
        if 'wgrad' in label:
            byts = 4.0 * (M_ * C_ + M_ * N_ + taps_ * C_ * N_)
        gbs = byts / (avg_us * 1e-6) / 1e9
        rows.append(dict(layer=label, kernel=kern, M=M_, N=N_, C=C_, taps=taps_, nphase=nph_, split=int(split),
                         launches_per_iter=round(n / steps, 2), avg_us=round(avg_us, 1), ms_per_iter=round(us / steps / 1e3, 3),
                         gflop=round(gf / n, 3), tflops=round(tf, 1),
                         executed_mfma_tflops=round(tf * (6 if limb else 1), 1), frac_mfma_peak=round(tf * (6 if limb else 1) / peak, 3),
                         algorithmic_GBps=round(gbs, 0), frac_hbm_peak=round(gbs / PEAK_HBM, 3)))
    rows.sort(key=lambda r: -r['ms_per_iter'])
    w = csv.DictWriter(open(out, 'w', newline='') if out else sys.stdout, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
    tot_ms = sum(r['ms_per_iter'] for r in rows)
    print(f'# {len(rows)} (layer, pass, shape) rows, {tot_ms:.2f} ms of GEMM-shaped kernels per iteration', file=sys.stderr)


if __name__ == '__main__':
    main()
