"""profiles/rNN_util.json from the two SQ counter passes of the round (mechanical: no hand-copied numbers): how busy the LDS arrays
and the matrix pipes of the chip are inside each CNN kernel -- the third and fourth roof next to HBM bytes and issued matrix FLOPs.

usage: python profiles/make_util.py profiles/r05_bench_pmc_lds.txt profiles/r05_bench_pmc_lds2.txt > profiles/r05_util.json

Counters (rocprofv3 --pmc ... --kernel-trace on `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline`, summarised
per launch by profiles/summarize_pmc.py; 512 agents per launch):
  SQ_LDS_IDX_ACTIVE        LDS-array cycles, summed over the 256 CUs  (/opt/skills/guides/MI355X_MICROARCH.md, LDS section)
  SQ_LDS_BANK_CONFLICT     the part of them added by bank conflicts
  SQ_VALU_MFMA_BUSY_CYCLES matrix-pipe cycles, summed over the 1024 SIMDs (= SQ_INSTS_MFMA x 32 for the 32x32x16 shape: checked below)
  GRBM_GUI_ACTIVE          the launch's cycles, summed over the 8 XCDs
busy fraction = (counter / units) / (GRBM_GUI_ACTIVE / 8).  The LDS peak the guide gives is 256 B / clk / CU for ds_read_b64 / b128
(157 TB/s at 2.4 GHz): `lds_GBps_equiv` = busy fraction x that peak at the launch's own clock."""
import json
import sys

NCU, NSIMD, NXCD = 256, 1024, 8
KERNELS = [
    ('conv1b_kernel<true> (fused crop -> conv1)', 'conv1b_kernel<true, 0>'),
    ('conv_ws_kernel<conv2>', 'conv_ws_kernel<Cin=16,Cout=32'),
    ('conv_wsx_kernel<conv3>', 'conv_wsx_kernel<Cin=32,Cout=64'),
    ('conv_wsx_kernel<conv4>', 'conv_wsx_kernel<Cin=64,Cout=64'),
    ('conv_bf6_kernel<conv3>', 'conv_bf6_kernel<Cin=32,Cout=64'),
    ('conv_bf6_kernel<conv4>', 'conv_bf6_kernel<Cin=64,Cout=64'),
    ('cnn_tail_kernel (conv5 + conv6 + Linear)', 'cnn_tail_kernel<'),
    ('scn::scene_bwd_sweep_kernel', 'scn::scene_bwd_sweep_kernel<false>'),
    ('scn::scene_fwd_step_kernel', 'scn::scene_fwd_step_kernel<false>'),
]


def rows(path):
    lines = [l.rstrip('\n') for l in open(path) if l.strip() and not l.startswith('/opt')]
    hdr = lines[0].split()[2:]
    out = {}
    for l in lines[1:]:
        name = l[:50].strip()
        vals = l[50:].split()
        if len(vals) != len(hdr) + 1:
            continue
        out[name] = dict(zip(hdr, (float(v) for v in vals[1:])))
    return out


def main():
    a, b = rows(sys.argv[1]), rows(sys.argv[2])
    res = {'_how': __doc__.split('\n\n')[0].replace('\n', ' ')}
    for name, prefix in KERNELS:
        ra = next((v for k, v in a.items() if k.startswith(prefix[:48])), None)
        rb = next((v for k, v in b.items() if k.startswith(prefix[:48])), None)
        if ra is None or rb is None:
            continue
        cyc = rb['GRBM_GUI_ACTIVE'] / NXCD
        lds = ra['SQ_LDS_IDX_ACTIVE'] / NCU / cyc
        mfma = rb['SQ_VALU_MFMA_BUSY_CYCLES'] / NSIMD / cyc
        res[name] = {
            'cycles_per_launch': int(cyc),
            'lds_array_busy_frac': round(lds, 4),
            'lds_bank_conflict_share_of_array_cycles': round(ra['SQ_LDS_BANK_CONFLICT'] / max(ra['SQ_LDS_IDX_ACTIVE'], 1.0), 4),
            'lds_cycles_per_instruction': round(ra['SQ_LDS_IDX_ACTIVE'] / max(ra['SQ_INSTS_LDS'], 1.0), 2),
            'lds_bytes_equiv_per_launch': int(ra['SQ_LDS_IDX_ACTIVE'] * 256),
            'mfma_pipe_busy_frac': round(mfma, 4),
            'mfma_cycles_per_instruction': round(rb['SQ_VALU_MFMA_BUSY_CYCLES'] / max(rb['SQ_INSTS_MFMA'], 1.0), 2),
            'lds_issue_stall_share_of_wave_cycles': round(ra['SQ_WAIT_INST_LDS'] / max(ra['SQ_WAVE_CYCLES'], 1.0), 4),
        }
    json.dump(res, sys.stdout, indent=1)
    sys.stdout.write('\n')


if __name__ == '__main__':
    main()
