#!/usr/bin/env python
"""Per-kernel averages of every counter collected by tools/gpu_pmc.sh (one rocprofv3 --pmc pass per group).
usage: tools/pmc_table.py <dir with group_*.csv>"""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    m = re.match(r'(?:void )?(?:eg::)?([A-Za-z0-9_]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main():
    d = sys.argv[1]
    table = defaultdict(OrderedDict)
    counters = []
    for f in sorted(glob.glob(os.path.join(d, 'group_*.csv'))):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                a = agg[short(r['Kernel_Name'])][r['Counter_Name']]
                a[0] += 1
                a[1] += float(r['Counter_Value'])
        for k, cs in agg.items():
            for cn, (n, s) in cs.items():
                if cn not in counters:
                    counters.append(cn)
                table[k][cn] = s / n
                table[k]['launches'] = n
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'launches'] + counters)
    for k, cs in sorted(table.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0) * kv[1].get('launches', 0)):
        w.writerow([k, cs.get('launches', '')] + [round(cs[c], 1) if c in cs else '' for c in counters])


if __name__ == '__main__':
    main()
