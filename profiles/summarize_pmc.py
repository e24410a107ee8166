#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel: mean counter value per dispatch."""
import csv
import re
import sys
from collections import defaultdict


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
        return 'conv<Cin=%s,Cout=%s,k=%s,in=%s,out=%s>' % m.groups()
    return name[:48]


def main(path, only=None):
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row['Kernel_Name'])
            if only and only not in k:
                continue
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            disp[k].add(row['Dispatch_Id'])
    counters = sorted({c for v in acc.values() for c in v})
    print('%-50s %6s ' % ('kernel', 'calls') + ' '.join('%22s' % c for c in counters))
    for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
        n = len(disp[k])
        print('%-50s %6d ' % (k, n) + ' '.join('%22.0f' % (acc[k][c] / n) for c in counters))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
