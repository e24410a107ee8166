"""Host-side logic that needs no GPU: scene collation, clique validation, synthetic generators, slot layout of
the collision pairs, sharding and the gloo scalar all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from strive_amd import synth, ops
from strive_amd.graph import Batch, clique_edge_index
from strive_amd.distributed import shard_scenes, GlobalMean


def test_clique_edge_index_matches_reference_order():
    ei = clique_edge_index(3)
    assert ei.tolist() == [[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]]
    assert clique_edge_index(1).shape == (2, 0)


def test_batch_collation_and_split():
    b, mi = synth.make_batch([3, 1, 4], key='hl')
    assert b.ptr.tolist() == [0, 3, 4, 8] and b.batch.tolist() == [0, 0, 0, 1, 2, 2, 2, 2]
    assert b.edge_index.shape[1] == 6 + 0 + 12 and int(b.edge_index.max()) == 7
    parts = b.to_data_list()
    assert [p.past.shape[0] for p in parts] == [3, 1, 4]
    assert parts[2].edge_index.tolist() == clique_edge_index(4).tolist()
    assert 'future' in b and 'nothing' not in b


def test_scene_info_accepts_cliques_and_rejects_other_graphs():
    b, _ = synth.make_batch([3, 2], key='hl2')
    info = ops.scene_info(b)
    assert info.P == 9 + 4 and info.pair_off.tolist() == [0, 3, 6, 9, 11]
    b2, _ = synth.make_batch([3, 2], key='hl2')
    b2.edge_index = b2.edge_index[:, :-1]
    with pytest.raises(NotImplementedError):
        ops.scene_info(b2)


def test_counter_generator_is_stable():
    a = synth.counter_uniform((4,), 'key')
    np.testing.assert_allclose(a, synth.counter_uniform((4,), 'key'))
    assert abs(float(a[0]) - 0.0) <= 1.0 and len(set(np.round(a, 12))) == 4
    # pinned values: a change here invalidates every golden fixture
    np.testing.assert_allclose(synth.counter_uniform((3,), 'pin'), [0.29416299043974026, 0.38165813672071747, 0.9328585409599534],
                               rtol=0, atol=1e-9)


def test_linspace5_matches_torch_linspace():
    from strive_amd.losses.adv_gen_nusc import _linspace5
    lo = synth.f32(synth.counter_uniform((64,), 'lo', -3.0, -0.5))
    hi = synth.f32(synth.counter_uniform((64,), 'hi', 0.5, 3.0))
    want = torch.stack([torch.linspace(lo[i].item(), hi[i].item(), 5) for i in range(64)], dim=0)
    assert torch.equal(_linspace5(lo, hi), want)


def test_collate_tgt_other_z():
    from strive_amd.utils.adv_gen_optim import collate_tgt_other_z
    b, _ = synth.make_batch([3, 1, 2], key='hl3')
    tgt = torch.arange(3, dtype=torch.float32).view(3, 1) + 100
    oth = torch.arange(3, dtype=torch.float32).view(3, 1)
    assert collate_tgt_other_z(b, tgt, oth).view(-1).tolist() == [100, 0, 1, 101, 102, 2]


def test_shard_scenes_balances_agents():
    sizes = [16] * 32
    parts = shard_scenes(sizes, 8)
    assert sorted(sum(parts, [])) == list(range(32)) and all(len(p) == 4 for p in parts)
    sizes = [2, 30, 7, 7, 12, 5, 19, 3, 3, 9]
    parts = shard_scenes(sizes, 4)
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert sorted(sum(parts, [])) == list(range(10)) and max(loads) - min(loads) <= max(sizes)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sizes = [5, 3, 8, 2, 6, 4]
    mine = shard_scenes(sizes, world)[rank]
    # every rank holds the penalties of its own scenes; the global mean must equal the single-batch mean
    vals = torch.cat([torch.arange(sizes[i], dtype=torch.float64) + 10 * i for i in mine]).requires_grad_(True)
    gm = GlobalMean()(vals)
    gm.backward()
    q.put((rank, float(gm), vals.grad.tolist(), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_global_mean_over_two_ranks_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [5, 3, 8, 2, 6, 4]
    allv = np.concatenate([np.arange(n) + 10 * i for i, n in enumerate(sizes)])
    covered = []
    for rank, val, grad, mine in res:
        assert abs(val - allv.mean()) < 1e-12
        assert all(abs(g - 1.0 / len(allv)) < 1e-15 for g in grad)
        covered += mine
    assert sorted(covered) == list(range(6))
