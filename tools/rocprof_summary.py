#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats CSV kept under profiles/.
usage: tools/rocprof_summary.py <results.db> <out.csv> [steps]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                            'from kernels group by name order by sum(duration) desc'))
    total = sum(r[2] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage', 'MsPerStep'])
        for n, c, s, a, mn, mx in rows:
            w.writerow([n, c, int(s), round(a, 1), int(mn), int(mx), round(100.0 * s / total, 3),
                        round(s / 1e6 / steps, 4)])
        w.writerow(['TOTAL', sum(r[1] for r in rows), int(total), '', '', '', 100.0, round(total / 1e6 / steps, 4)])


if __name__ == '__main__':
    main()
