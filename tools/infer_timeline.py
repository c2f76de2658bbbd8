#!/usr/bin/env python
"""Launch-by-launch timeline of the LAST inference step in a rocprofv3 (rocpd sqlite) kernel trace of
`bench.py --mode infer`: for every launch its queue, start offset, duration, and the gap to the previous kernel's end on the
same queue and on the whole device -- gaps (launch latency / dependency stalls) versus tails (kernel durations) is what decides
between graph capture and kernel fusion.
usage: tools/infer_timeline.py <results.db> <launches_per_step, 0 = detect> [steps_to_average]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db, per = sys.argv[1], int(sys.argv[2])
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = 'select name, start, end%s from kernels order by start' % (', ' + qcol if qcol else '')
    rows = list(cur.execute(sel))
    print('columns:', cols)
    print('launches in trace:', len(rows))
    if per <= 0:     # find the period of the launch sequence at the end of the trace
        names = [r[0] for r in rows]
        for cand in range(8, len(names) // 3):
            if names[-cand:] == names[-2 * cand:-cand] == names[-3 * cand:-2 * cand]:
                per = cand
                break
        print('detected launches per step:', per)
    tail = rows[-per * nsteps:]
    # aggregate over the last nsteps steps by position in the step
    agg = defaultdict(lambda: [0.0, 0.0, 0.0, None, None])
    span_tot = busy_tot = 0.0
    for s in range(nsteps):
        step = tail[s * per:(s + 1) * per]
        t0 = step[0][1]
        last_q = {}
        last_dev = None
        ev = []
        for i, r in enumerate(step):
            name, st, en = r[0], r[1], r[2]
            q = r[3] if qcol else 0
            gq = (st - last_q[q]) if q in last_q else 0
            gd = (st - last_dev) if last_dev is not None else 0
            a = agg[i]
            a[0] += (en - st) / nsteps
            a[1] += gq / nsteps
            a[2] += gd / nsteps
            a[3], a[4] = name, q
            last_q[q] = en
            last_dev = en if last_dev is None else max(last_dev, en)
            ev += [(st, 1), (en, -1)]
        ev.sort()
        depth, last, busy = 0, t0, 0
        for t, d in ev:
            if depth > 0:
                busy += t - last
            last = t
            depth += d
        span_tot += (max(r[2] for r in step) - t0) / nsteps
        busy_tot += busy / nsteps
    print('step span %.1f us, >=1 kernel running %.1f us, idle %.1f us, sum of durations %.1f us (avg of the last %d steps)' %
          (span_tot / 1e3, busy_tot / 1e3, (span_tot - busy_tot) / 1e3, sum(a[0] for a in agg.values()) / 1e3, nsteps))
    print('%3s %-5s %8s %9s %9s  %s' % ('#', 'queue', 'dur_us', 'gap_q_us', 'gap_dev', 'kernel'))
    for i in range(per):
        a = agg[i]
        print('%3d %-5s %8.1f %9.1f %9.1f  %s' % (i, a[4], a[0] / 1e3, a[1] / 1e3, a[2] / 1e3, (a[3] or '')[:150]))


if __name__ == '__main__':
    main()
