"""Multi-GPU: scene-sharded replicas.

Edges never cross scenes (reference src/datasets/nuscenes_dataset.py:678-687 builds per-scene cliques and PyG only
offsets indices), latents and Adam state are per agent, and every normaliser is per sample -- so the optimisation
loops shard embarrassingly: each rank (one process per GPU) owns whole scenes and runs the unchanged loop on them;
the raster is replicated.  There is no data-path collective.

The one cross-scene coupling in the reference losses is that every term is a batch-wide ``.mean()`` over compacted
lists (reference src/losses/adv_gen_nusc.py:229-250, 317-336), i.e. the *normalisation* of the gradient depends on
the batch composition.  Running sharded is exactly the reference run with a smaller ``batch_size``
(src/adv_scenario_gen.py:237) and is what bench.py measures ("scaling": "weak").  ``GlobalMean`` is the optional
few-scalar all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) that reproduces the single-batch
normalisation across ranks when bit-for-bit batch semantics are wanted.
"""
import torch
import torch.distributed as dist


def shard_scenes(sizes, world_size):
    """Greedy longest-processing-time partition of scenes over ranks, balancing sum(n) (the per-agent map CNN
    dominates the cost) with sum(n^2) (edges / collision pairs) as the tie break.  Returns a list of scene-index
    lists, one per rank; deterministic."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [[0, 0, r] for r in range(world_size)]
    out = [[] for _ in range(world_size)]
    for i in order:
        loads.sort(key=lambda t: (t[0], t[1], t[2]))
        n = int(sizes[i])
        loads[0][0] += n
        loads[0][1] += n * n
        out[loads[0][2]].append(i)
    for r in range(world_size):
        out[r].sort()
    return out


class GlobalMean(object):
    """mean over the union of per-rank lists:  all_reduce([sum, count]) -> sum/count, differentiable w.r.t. the
    local entries (d/dx_i = 1/global_count)."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, values, count=None):
        n = float(values.numel()) if count is None else float(count)
        local_sum = values.sum()
        stat = torch.stack([local_sum.detach(), torch.tensor(n, device=values.device, dtype=local_sum.dtype)])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=self.group)
        gcount = torch.clamp(stat[1], min=1.0)
        # value = global mean; gradient flows through the local sum only
        return (local_sum - local_sum.detach() + stat[0]) / gcount
