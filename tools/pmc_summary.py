#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Units: the counters are KiB; on gfx950 FETCH_SIZE
reads exactly 1/2 of the bytes of a wide coalesced stream, so the read side is doubled (guide, HBM section).
usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv>"""
import csv
import sys
from collections import defaultdict


def load(path):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            a = agg[r['Kernel_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
            a[2] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return agg


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    rows = []
    for k in fe:
        n, fsum, tsum = fe[k]
        wsum = wr.get(k, [0, 0.0, 0.0])[1]
        rd = 2.0 * fsum * 1024 / n          # gfx950 correction: x2 on the read side
        wb = wsum * 1024 / max(wr.get(k, [1])[0], 1)
        rows.append((k, n, rd, wb, rd + wb, tsum / n))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(sys.argv[3], 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel', 'Launches', 'ReadBytesPerLaunch(2xFETCH_SIZE)', 'WriteBytesPerLaunch', 'HbmBytesPerLaunch',
                    'AvgNsUnderPmc', 'GBps'])
        for k, n, rd, wb, tot, ns in rows:
            w.writerow([k, n, int(rd), int(wb), int(tot), int(ns), round(tot / ns, 1) if ns else ''])


if __name__ == '__main__':
    main()
