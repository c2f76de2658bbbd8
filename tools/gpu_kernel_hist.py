#!/usr/bin/env python
"""Per-kernel duration list from a rocprofv3 rocpd database, with the grid size: tools/gpu_kernel_hist.py <db> <substr>"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = 'grid_x' if 'grid_x' in cols else None
q = 'select name, duration' + (', grid_x, grid_y, workgroup_x, lds_size' if gx else '') + ' from kernels where name like ?'
agg = {}
for r in cur.execute(q, ('%' + pat + '%',)):
    key = (r[0][:60],) + tuple(r[2:])
    a = agg.setdefault(key, [0, 0.0, 1e18])
    a[0] += 1
    a[1] += r[1]
    a[2] = min(a[2], r[1])
for k, (n, tot, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, 'calls', n, 'avg_us %.1f' % (tot / n / 1e3), 'min_us %.1f' % (mn / 1e3))
