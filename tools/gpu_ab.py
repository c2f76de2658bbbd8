"""A/B two builds of the library in one process-group on one box: EG_LIB_PATH selects the .so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]      # further bench.py arguments, e.g. --dtype bf16
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, EG_LIB_PATH=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2',
                              '--no-cpu-baseline', '--profile-steps', '0'] + extra, env=env, capture_output=True, text=True).stdout
        import json
        res[l].append(json.loads(out.strip().splitlines()[-1])['value'])
for l in libs:
    print(l, ['%.1f' % v for v in res[l]], 'median %.1f' % sorted(res[l])[len(res[l]) // 2])
