"""Runs every GPU parity check without stopping at failures and prints a table (for gpurun)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np   # noqa: E402
import edgegan_amd as ea  # noqa: E402
import parity  # noqa: E402


def main():
    which = sys.argv[1:] or ['ops', 'nets', 'substeps']
    B = int(os.environ.get('EG_DIAG_B', '2'))
    cfg = ea.Config(batch_size=B, num_classes=int(os.environ.get('EG_DIAG_NC', '14')))
    values = ea.init_variables(cfg, int(os.environ.get('EG_DIAG_SEED', '0')))
    net = ea.EdgeGAN(cfg, values)
    t0 = time.time()
    for stage, fn in (('ops', lambda: parity.all_op_checks(net)), ('sweep', lambda: parity.batch_sweep_checks(net)), ('real', lambda: parity.real_shape_checks(net)), ('nets', lambda: parity.net_checks(net, values)),
                      ('substeps', lambda: parity.substep_checks(net, values, ea.synthetic_batch(cfg, seed=int(os.environ.get('EG_DIAG_BSEED', '1234')))))):
        if stage not in which:
            continue
        try:
            rows = fn()
            nfail = sum(1 for _, e, t in rows if not e <= t)
            print(f'==== {stage}: {len(rows)} checks, {nfail} failed, {time.time() - t0:.1f}s')
            print(parity.format_rows(rows, only_fail=False))
        except Exception:
            print(f'==== {stage}: EXCEPTION')
            traceback.print_exc()
        sys.stdout.flush()


if __name__ == '__main__':
    main()
