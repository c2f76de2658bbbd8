#!/usr/bin/env python
"""How much of an iteration's kernel time overlaps?  Reads two rocprofv3 (rocpd sqlite) kernel traces of the same bench command --
lanes on (default) and EG_LANES=0 (serialised) -- and prints, for the concurrent run: the span of the traced window, the time at
least one kernel was running (union), the idle gaps, the time-weighted number of kernels in flight, and per kernel name the
average duration next to the serialised run's (how much a kernel is stretched by what runs beside it).
Only the last <window_ms> of the concurrent trace are analysed (the timed steps of bench.py: steps x ms_per_step).
usage: tools/trace_overlap.py <concurrent.db> <serial.db> [top] [window_ms]"""
import sqlite3
import sys
from collections import defaultdict


def load(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    rows = list(cur.execute('select name, start, end from kernels order by start'))
    return cols, rows


def load_fill(db):
    """(start, end, fill): fill = the fraction of the 256 CUs' block slots a launch can occupy at once, from its grid and what
    one CU holds of it (LDS 160 KB, 512 VGPRs per SIMD lane, 8 waves per SIMD of this image's kernels' sizes)."""
    cur = sqlite3.connect(db).cursor()
    out = []
    for n, s, e, gx, gy, gz, wx, wy, wz, lds, vg, ag in cur.execute(
            'select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, lds_size, vgpr_count, accum_vgpr_count '
            'from kernels order by start'):
        wg = max(1, wx * wy * wz)
        blocks = max(1, (gx * gy * gz) // wg)
        waves = (wg + 63) // 64
        regs = max(8, ((vg or 0) + (ag or 0) + 7) // 8 * 8)
        per_simd = max(1, min(8, 512 // regs))
        by_regs = max(1, per_simd * 4 // waves)
        by_lds = max(1, (160 * 1024) // lds) if lds else 64
        slots = 256 * min(by_regs, by_lds, max(1, 32 // waves))
        out.append((n, s, e, min(1.0, blocks / slots), blocks, slots // 256))
    return out


def main():
    cols, conc = load(sys.argv[1])
    _, ser = load(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    print('columns of the kernels view:', cols)
    t0, t1 = conc[0][1], max(r[2] for r in conc)
    if len(sys.argv) > 4:
        t0 = t1 - int(float(sys.argv[4]) * 1e6)
        conc = [(n, max(s, t0), e) for n, s, e in conc if e > t0]
    ev = []
    for _, s, e in conc:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy, hist = 0, t0, 0, defaultdict(int)
    for t, d in ev:
        if depth > 0:
            busy += t - last
        hist[min(depth, 6)] += t - last
        last = t
        depth += d
    span = t1 - t0
    print('span %.2f ms, >= 1 kernel running %.2f ms (%.1f %%), idle %.2f ms; sum of kernel durations %.2f ms (%.2f x span)' % (
        span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, sum(e - s for _, s, e in conc) / 1e6,
        sum(e - s for _, s, e in conc) / span))
    print('time by kernels in flight:', {k: '%.1f %%' % (100.0 * v / span) for k, v in sorted(hist.items())})
    # block-slot fill: integrate min(1, sum of the fills of the kernels in flight)
    fl = load_fill(sys.argv[1])
    fl = [(n, max(s, t0), e, f, b, k) for n, s, e, f, b, k in fl if e > t0]
    ev2 = []
    for n, s, e, f, b, k in fl:
        ev2.append((s, f))
        ev2.append((e, -f))
    ev2.sort()
    cur_f, last, filled, low = 0.0, t0, 0.0, defaultdict(float)
    for t, d in ev2:
        filled += min(1.0, max(0.0, cur_f)) * (t - last)
        bucket = min(4, int(max(0.0, min(cur_f, 0.999)) * 4))
        low[bucket] += t - last
        last = t
        cur_f += d
    print('block-slot fill (launch grids against resident-block capacity): %.1f %% of the span; time by fill quartile %s' % (
        100.0 * filled / span, {('%d-%d %%' % (25 * k, 25 * k + 25)): '%.1f %%' % (100.0 * v / span) for k, v in sorted(low.items())}))
    small = defaultdict(lambda: [0.0, 0])
    for n, s, e, f, b, k in fl:
        if f < 0.5:
            small[n][0] += (e - s)
            small[n][1] += 1
    print('launches that fill < 50 %% of the slots, by total time:')
    for n, (tot, c) in sorted(small.items(), key=lambda kv: -kv[1][0])[:15]:
        print('   %-100s %6d launches %8.2f ms' % (n[:100], c, tot / 1e6))
    # solo time: how long each kernel ran with NOTHING beside it -- the part of the span that only a shorter chain (fewer or
    # faster dependent launches) can shrink, whatever the other lanes do
    bounds = sorted(set([t for t, _ in ev]))
    import bisect
    starts = sorted((s, e, n) for n, s, e in conc)
    solo = defaultdict(lambda: [0.0, 0])
    active = []
    j = 0
    for a, b in zip(bounds, bounds[1:]):
        while j < len(starts) and starts[j][0] <= a:
            active.append(starts[j])
            j += 1
        active = [k for k in active if k[1] > a]
        if len(active) == 1:
            solo[active[0][2]][0] += b - a
    for n, s, e in conc:
        solo[n][1] += 1
    tot_solo = sum(v[0] for v in solo.values())
    print('solo time (exactly one kernel in flight): %.2f ms = %.1f %% of the span; by kernel:' % (tot_solo / 1e6, 100.0 * tot_solo / span))
    for n, (tt, c) in sorted(solo.items(), key=lambda kv: -kv[1][0])[:top]:
        print('   %-100s %6d launches %8.2f ms solo' % (n[:100], c, tt / 1e6))
    sd = defaultdict(lambda: [0, 0])
    for n, s, e in ser:
        sd[n][0] += e - s
        sd[n][1] += 1
    cd = defaultdict(lambda: [0, 0])
    for n, s, e in conc:
        cd[n][0] += e - s
        cd[n][1] += 1
    print('%-90s %8s %10s %10s %7s' % ('kernel', 'calls', 'conc us', 'serial us', 'stretch'))
    for n, (tot, c) in sorted(cd.items(), key=lambda kv: -kv[1][0])[:top]:
        s = sd.get(n)
        sa = s[0] / s[1] / 1e3 if s and s[1] else float('nan')
        print('%-90s %8d %10.1f %10.1f %7.2f' % (n[:90], c, tot / c / 1e3, sa, tot / c / 1e3 / sa if sa == sa else float('nan')))


if __name__ == '__main__':
    main()
