#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (default output format of ROCm 7.2's rocprofv3 --kernel-trace --stats)
into the per-kernel statistics table that --stats prints: calls, total / average / min / max duration, share."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void\s+', '', name)
    m = re.match(r'conv_bf6_kernel<BfCfg<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        return 'conv_bf6_kernel<Cin=%s,Cout=%s,k=%s,in=%s,out=%s>' % m.groups()
    m = re.match(r'conv_wsx_kernel<BfCfg<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        return 'conv_wsx_kernel<Cin=%s,Cout=%s,k=%s,in=%s,out=%s>' % m.groups()
    m = re.match(r'conv_ws_kernel<BfCfg<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        return 'conv_ws_kernel<Cin=%s,Cout=%s,k=%s,in=%s,out=%s>' % m.groups()
    m = re.match(r'conv_bf6s_kernel<BfsCfg<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        return 'conv_bf6s_kernel<Cin=%s,Cout=%s,in=%s,out=%s,S=%s>' % m.groups()
    m = re.match(r'conv_mfma_kernel<ConvCfg<(\d+), (\d+), (\d+), (\d+), (\d+)', name)
    if m:
        return 'conv_mfma_kernel<Cin=%s,Cout=%s,k=%s,in=%s,out=%s>' % m.groups()
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, (end - start) from kernels').fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print('%-88s %7s %12s %11s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-88s %7d %12.1f %11.2f %10.2f %10.2f %6.2f' % (k, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                              100.0 * a[1] / total))
    print('total kernel time: %.3f ms over %d dispatches' % (total / 1e6, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1])
