#!/usr/bin/env python3
"""Idle time between consecutive kernels in a rocprofv3 rocpd database: where does the GPU wait for the host?
usage: gap_report.py results.db [min_gap_us] [window_kernel skip]
With `window_kernel skip` only the kernels from the (skip+1)-th launch of `window_kernel` on are analysed (e.g.
`rollout_init_kernel 2` = the timed closures of bench.py --warmup 2)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(.*$', '', n)
    n = re.sub(r'^void\s+', '', n)
    return re.sub(r'<.*$', '', n)[:36]


def main(path, min_gap=5.0, window=None, skip=0):
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    if window:
        hits = [i for i, r in enumerate(rows) if window in r[0]]
        rows = rows[hits[skip]:]
        print('window: from launch %d of %s (%d launches of it inside)' % (skip + 1, window, len(hits) - skip))
    busy = sum(e - s for _, s, e in rows) / 1e3
    span = (rows[-1][2] - rows[0][1]) / 1e3
    gaps = defaultdict(lambda: [0, 0.0])
    last_end = rows[0][2]
    idle = 0.0
    for i in range(1, len(rows)):
        n, s, e = rows[i]
        g = (s - last_end) / 1e3
        if g > 0:
            idle += g
            if g >= min_gap:
                k = '%s -> %s' % (short(rows[i - 1][0]), short(n))
                gaps[k][0] += 1
                gaps[k][1] += g
        last_end = max(last_end, e)
    print('kernels %d, span %.1f ms, busy (sum of durations) %.1f ms, idle between kernels %.1f ms' % (len(rows), span / 1e3, busy / 1e3, idle / 1e3))
    print('%-76s %6s %10s %9s' % ('gap >= %.0f us between' % min_gap, 'count', 'total_us', 'avg_us'))
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%-76s %6d %10.1f %9.1f' % (k, c, t, t / c))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0, sys.argv[3] if len(sys.argv) > 3 else None,
         int(sys.argv[4]) if len(sys.argv) > 4 else 0)
