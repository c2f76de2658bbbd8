"""Pretty-print a bench.py JSON line (file argument): headline + per-kernel table."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.pop('roofline')
print(d['dtype'], d['value'], 'img/s', d['ms_per_step'], 'ms/step', r['all_gemm_kernels'])
print('dominant', r['kernel'], r['achieved'], '/', r['peak'], r['unit'], 'frac', r['frac'], 'traffic', r['traffic'])
for k, v in sorted(r['per_kernel'].items(), key=lambda kv: -kv[1]['ms_per_step']):
    print(f'  {k:28s} {v}')
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline'])
