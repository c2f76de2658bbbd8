"""A/B two ENVIRONMENT settings of the same build in one process group on one box:
   python tools/gpu_ab_env.py EG_LANES=7 EG_LANES=15 [rounds] [extra bench args...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {a: [], b: []}
for r in range(rounds):
    for s in (a, b):
        env = dict(os.environ)
        for kv in s.split(','):
            k, v = kv.split('=')
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu-baseline',
                              '--no-alt', '--profile-steps', '0'] + extra, env=env, capture_output=True, text=True).stdout
        res[s].append(json.loads(out.strip().splitlines()[-1])['value'])
for s in (a, b):
    print(s, ['%.1f' % v for v in res[s]], 'median %.1f' % sorted(res[s])[len(res[s]) // 2])
