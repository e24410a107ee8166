"""Phase ticks of the scene-resident rollout kernels (csrc/scene_rollout.h): STRIVE_SCENE_PROF=1 makes workgroup 0 add the
core-clock ticks of every barrier-delimited phase to counters at the start of the rollout workspace.
usage: python tools/scene_phase_probe.py [scenes agents FT]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ['STRIVE_SCENE_PROF'] = '1'
from strive_amd import synth, ops  # noqa: E402
from util import product_model  # noqa: E402

FWD = ['features+mlp_in', 'P/Q', 'edge layer0', 'edge LN0+split', 'edge mma1', 'edge LN1+split', 'edge mma2', 'edge max',
       'tail of edges', 'update+mlp_out', 'bike', 'GRU']
BWD = ['GRU', 'bike+loads', 'mlp_out bwd', 'update bwd', 'edge setup+split', 'edge mma2', 'edge LN1 bwd', 'edge mma1', 'edge LN0 bwd',
       'dP/dQ/drel', 'rel_pose+gpos', 'node1 P/Q', 'mlp_in bwd']


def main():
    B, n, FT = (int(v) for v in (sys.argv[1:4] + ['32', '16', '16'][len(sys.argv) - 1:]))
    dev = torch.device('cuda', 0)
    m, _ = product_model(device=dev)
    raster, dx = synth.make_raster(2048, 2048)
    env = synth.SyntheticMapEnv(raster, dx).to(dev)
    batch, map_idx = synth.make_batch([n] * B, key='probe', map_extent=(400.0, 400.0))
    g, mi = batch.to(dev), map_idx.to(dev)
    with torch.no_grad():
        emb = m.embed(g, mi, env)
    z = emb['prior_out'][0].clone().requires_grad_(True)
    reps = 5
    for it in range(reps + 1):
        if it == 1:
            for buf in ops._ws_cache.values():
                buf.zero_()
        pred = m.decode_embedding(z, emb, g, mi, env, nfuture=FT)['future_pred']
        pred.square().sum().backward()
    torch.cuda.synchronize()
    ws = [b for k, b in ops._ws_cache.items() if k[1] == 'rollout'][0]
    ticks = ws[:64 * 8].view(torch.int64).cpu().tolist()
    print('%d scenes x %d agents, FT %d: ticks of workgroup 0 per rollout (mean of %d), us at 2.1 GHz' % (B, n, FT, reps))
    for name, off, n_ in (('forward (sum over %d steps)' % FT, 0, FWD), ('backward sweep', 32, BWD)):
        tot = sum(ticks[off:off + len(n_)]) / reps
        print('%s: %.0f ticks = %.1f us' % (name, tot, tot / 2100.0))
        for i, nm in enumerate(n_):
            v = ticks[off + i] / reps
            print('   %-20s %10.0f  %7.1f us  %5.1f %%' % (nm, v, v / 2100.0, 100.0 * v / max(tot, 1.0)))


if __name__ == '__main__':
    main()
