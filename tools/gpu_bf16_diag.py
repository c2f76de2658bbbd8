"""bf16 compute mode: op parity against the bf16-rounded oracle, then network / sub-step deviation from the fp64
oracle (printed as distributions so the test tolerances can be set from measurements)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import edgegan_amd as ea
import parity

what = sys.argv[1:] or ['ops', 'nets', 'steps']
if 'ops' in what:
    net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='bf16'))
    rows = parity.bf16_op_checks(net)
    bad = [r for r in rows if not r[1] <= r[2]]
    print(f'==== bf16 ops: {len(rows)} checks, {len(bad)} failed; worst {max(r[1] for r in rows):.2e}')
    print(parity.format_rows(bad[:40]))
    net.close()
if 'nets' in what:
    cfg = ea.Config(batch_size=2, compute_dtype='bf16')
    values = ea.init_variables(cfg, 1)
    net = ea.EdgeGAN(cfg, values)
    rows = parity.net_checks(net, values, bf16=True)
    print('==== bf16 nets vs bf16-operand oracle (max-norm relative error)')
    print(parity.format_rows(rows))
    net.close()
if 'steps' in what:
    for B, nc in ((2, 14), (4, 0))[:int(os.environ.get('NCFG', '2'))]:
        cfg = ea.Config(batch_size=B, num_classes=nc, compute_dtype='bf16')
        values = ea.init_variables(cfg, 0)
        net = ea.EdgeGAN(cfg, values)
        rows = parity.substep_checks(net, values, ea.synthetic_batch(cfg), grad_tol=1.0, bf16=True)
        g = [r for r in rows if ' grad ' in r[0] and 'structural' not in r[0]]
        l = [r for r in rows if ' loss ' in r[0]]
        o = [r for r in rows if r not in g and r not in l]
        errs = np.array([r[1] for r in g])
        print(f'==== bf16 sub-steps B={B} classes={nc}: {len(g)} gradient rows: median {np.median(errs):.2e} '
              f'p90 {np.percentile(errs, 90):.2e} max {errs.max():.2e}')
        for k in range(1, 8):
            e = np.array([r[1] for r in g if f'substep {k} ' in r[0]])
            if len(e):
                print(f'   substep {k}: n={len(e)} median {np.median(e):.2e} max {e.max():.2e}')
        print('   worst gradient rows:'); print(parity.format_rows(sorted(g, key=lambda r: -r[1])[:14]).replace('ok  ', '    '))
        print('   losses:'); print(parity.format_rows(l))
        bad = [r for r in o if not r[1] <= r[2]]
        print(f'   other rows: {len(o)}, failed {len(bad)}'); print(parity.format_rows(bad[:10]))
        net.close()
