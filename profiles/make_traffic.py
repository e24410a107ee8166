"""profiles/rNN_traffic.json from the rocprofv3 PMC summaries of the same round (mechanical: no hand-copied numbers).

usage: python profiles/make_traffic.py profiles/r02_bench_pmc_fetch_size.txt profiles/r02_bench_pmc_write_size.txt \
                                       profiles/r02_bench_kernel_stats.txt > profiles/r02_traffic.json

FETCH_SIZE / WRITE_SIZE are KB per launch (profiles/summarize_pmc.py prints the per-launch average).  Correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads
-> x2; WRITE_SIZE x1 (cross-check: conv2 writes 512 x 32 x 61 x 61 x 4 B = 243.9 MB, counter 244.1 MB)."""
import json
import sys

N = 512
# bench.py's names -> (rocprof kernel-name prefix, algorithmic bytes per agent: input once + output once)
KERNELS = [
    ('conv1b_kernel<true> (fused crop -> conv1)', 'conv1b_kernel<true, 0>', 4 * 256 * 256 + 16 * 125 * 125 * 4),
    ('conv_bf6_kernel<conv2>', 'conv_bf6_kernel<Cin=16,Cout=32', (16 * 125 * 125 + 32 * 61 * 61) * 4),
    ('conv_ws_kernel<conv2>', 'conv_ws_kernel<Cin=16,Cout=32', (16 * 125 * 125 + 32 * 61 * 61) * 4),
    ('conv_wsx_kernel<conv3>', 'conv_wsx_kernel<Cin=32,Cout=64', (32 * 61 * 61 + 64 * 29 * 29) * 4),
    ('conv_wsx_kernel<conv4>', 'conv_wsx_kernel<Cin=64,Cout=64', (64 * 29 * 29 + 64 * 14 * 14) * 4),
    ('conv_bf6_kernel<conv3>', 'conv_bf6_kernel<Cin=32,Cout=64', (32 * 61 * 61 + 64 * 29 * 29) * 4),
    ('conv_bf6_kernel<conv4>', 'conv_bf6_kernel<Cin=64,Cout=64', (64 * 29 * 29 + 64 * 14 * 14) * 4),
    ('cnn_tail_kernel (conv5 + conv6 + Linear)', 'cnn_tail_kernel<', (64 * 14 * 14 + 64) * 4),
]


def table(path, col):
    out = {}
    for line in open(path):
        parts = line.rstrip('\n').split()
        if len(parts) < 3 or parts[0] == 'kernel':
            continue
        try:
            calls, val = int(parts[-1 - col]), None
        except ValueError:
            continue
        out[line[:52].strip()] = line
    return out


def lookup(path, prefix, field):
    """value of the given trailing column (-1 = last) of the row whose kernel name starts with prefix (names are cut at 50)"""
    for line in open(path):
        if line.startswith(prefix[:48]):
            return float(line.split()[field])
    return None


def main():
    fetch, write, stats = sys.argv[1:4]
    res = {'_how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes, profiles/rNN_commands.sh of the round) on `python '
                   'bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline`, summarised by profiles/summarize_pmc.py and '
                   'turned into this file by profiles/make_traffic.py; KB per launch of 512 agents; FETCH_SIZE x2 on gfx950 '
                   '(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), WRITE_SIZE x1; avg_us = rocprofv3 kernel-trace average '
                   'of the same command (the kernel_stats file of the round).'}
    for name, prefix, alg in KERNELS:
        f = lookup(fetch, prefix, -1)
        w = lookup(write, prefix, -1)
        us = None
        for line in open(stats):
            if line.startswith(prefix[:48]):
                us = float(line.split()[-4])          # calls total avg min max %  -> avg
                break
        if f is None or w is None or us is None:
            continue
        rb, wb = int(f * 1024 * 2), int(w * 1024)
        tot = rb + wb
        res[name] = {'agents_per_launch': N, 'fetch_size_kb_raw': int(f), 'write_size_kb_raw': int(w), 'read_bytes_corrected': rb,
                     'write_bytes': wb, 'bytes_per_launch': tot, 'algorithmic_bytes_per_launch': alg * N,
                     'traffic_over_algorithmic': round(tot / float(alg * N), 3), 'avg_us_rocprof': us,
                     'achieved_GBps_measured_traffic': round(tot / us / 1e3, 1),
                     'frac_of_hbm_peak_8TBps': round(tot / us / 1e3 / 8000.0, 3)}
    json.dump(res, sys.stdout, indent=1)
    sys.stdout.write('\n')


if __name__ == '__main__':
    main()
