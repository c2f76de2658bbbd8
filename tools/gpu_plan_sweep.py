"""Sweep of stream -> hardware-queue placements (model.cpp: EG_STREAM_PLAN, experiments build) at two workloads."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = {'P7 a|b+leaves|c|d': 'bcdxAxxxBxxxCxxxD', 'P9 a|b|c+leaves|d': 'bcdxxAxxxBxxxCxxxD', 'P10 a|b|c|d+leaves': 'bcdxxxAxxxBxxxCxxxD',
         'P11 a|b+A,D|c+B,C|d': 'bcdxABxxDC', 'P13 a|b+A,C|c+B,D|d': 'bcdxABxxCD', 'P14 a|b+D|c+A,B,C|d': 'bcdxDAxxxBxxxC'}
if len(sys.argv) > 1:
    PLANS = {k: v for k, v in PLANS.items() if k.split()[0] in sys.argv[1:]}
def run(plan, extra, rounds):
    vals = []
    for _ in range(rounds):
        env = dict(os.environ, EG_STREAM_PLAN=plan, EG_LIB_PATH=os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so'))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-alt',
                              '--profile-steps', '0'] + extra, env=env, capture_output=True, text=True)
        try:
            vals.append(json.loads(out.stdout.strip().splitlines()[-1])['value'])
        except Exception:
            print('FAILED', plan, out.stderr[-800:])
            return None
    return sorted(vals)[len(vals) // 2]
for name, plan in PLANS.items():
    a = run(plan, [], 3)
    b = run(plan, ['--dtype', 'bf16'], 2)
    c = run(plan, ['--height', '256', '--width', '512', '--batch', '16', '--steps', '4', '--dtype', 'bf16'], 2)
    d = run(plan, ['--height', '256', '--width', '512', '--batch', '16', '--steps', '4'], 2)
    print('%-42s %-20s 64x128 f32x3 %7.1f  bf16 %7.1f | 256x512 bf16 %6.1f  f32x3 %6.1f' % (name, plan, a or 0, b or 0, c or 0, d or 0), flush=True)
