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


# ------------------------------------------------------------------------------------------------
# data-parallel training (BASELINE.json configs[3]: train_traffic.cfg over 8 x MI355X)
# ------------------------------------------------------------------------------------------------

class DataParallelTrainer(object):
    """One optimisation step of the reference's training loop (src/train_traffic.py:103-131) with the scenes of a batch
    sharded over the ranks -- one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box,
    "gloo" in the CPU tests).

    The reference's loss is a sum of batch-wide means with DIFFERENT denominators (reconstruction over visible frames,
    KL over agents, vehicle collisions over ordered in-scene pairs, environment collisions over ego frames:
    src/losses/traffic_model.py:55-105), so averaging per-rank gradients would not reproduce the single-process step.
    Here every rank forms its part of the GLOBAL loss -- local sums over global counts -- and the gradients are then
    simply summed:
      1. all_reduce(SUM) of the four data-dependent counts (they depend on the batch only, not on the forward);
      2. forward -> TrafficModelLoss -> backward of the rank's share of the global loss (a RuntimeError is caught like the
         reference's per-batch try/except);
      3. ONE all_reduce(SUM) of the flat gradient bucket (1.09 M fp32 = 4.37 MB for NC = 2): if any rank failed, every rank
         skips the optimiser step (the skip vote), otherwise ``optimizer.step()`` runs -- identical parameters on every rank.
         The bucket is persistent and every ``p.grad`` is a view into it, so nothing is copied in or out.
    The skip vote never touches the device (round 5): a failure IS a host event (the exception this rank's Python caught), so the
    ranks exchange their flags through a HOST collective -- one all_reduce of a one-element CPU tensor over a gloo group next to
    the RCCL group -- while the device queue keeps running; the bucket's last element still carries the flag for whoever inspects
    the reduced bucket, but nobody reads it.  (Rounds 3-4 read it back after the gradient all-reduce: one device->host
    synchronisation per step and rank as soon as N > 1.)
    """

    def __init__(self, model, loss_fn, optimizer, group=None):
        self.model, self.loss_fn, self.optimizer, self.group = model, loss_fn, optimizer, group
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.last_error = None
        # host-side group for the skip vote: the default group itself when it is gloo (CPU tests), else a gloo group over the same
        # ranks (created collectively: every rank constructs its trainer)
        self.vote_group, self.vote_transport, self._vote = None, 'single rank: no vote', False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self._vote = True
            if dist.get_backend(group) == 'gloo':
                self.vote_group, self.vote_transport = group, 'host collective (the gloo group itself)'
            else:
                ranks = dist.get_process_group_ranks(group) if group is not None else None
                import os
                if os.environ.get('MASTER_ADDR', '') in ('127.0.0.1', 'localhost'):
                    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')       # single-node jobs: the container's hostname may not resolve
                try:
                    self.vote_group = dist.new_group(ranks=ranks, backend='gloo')
                    self.vote_transport = 'host collective (gloo group beside %s)' % dist.get_backend(group)
                except Exception as e:       # no usable host transport: the vote falls back to the reduced bucket (one device read per step)
                    self.vote_group = None
                    self.vote_transport = 'device read of the reduced bucket (gloo group unavailable: %s)' % (str(e).splitlines()[0][:120],)
        # ONE persistent flat gradient buffer; every p.grad is a view into it (autograd accumulates in place), so the bucket
        # that goes through the all-reduce IS the gradients: no per-parameter copy in, no copy out.  The last element carries
        # the skip vote.
        dev = self.params[0].device
        self.bucket = torch.zeros((self.numel + 1,), dtype=torch.float32, device=dev)
        self._views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self._views.append(self.bucket[off:off + n].view_as(p))
            off += n

    def _bind_grads(self):
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _all_reduce(self, t):
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    @staticmethod
    def batch_counts(scene_graph, FT):
        """(visible frames, agents, ordered in-scene pairs, ego frames) of this rank's scenes: float64 on the graph's device.
        They depend on the batch only, so they are computed once per batch (kept on the graph object) and never read on the
        host: a read-back here made the host wait for the previous step's kernels at the start of every step."""
        key = (scene_graph.ptr.data_ptr(), scene_graph.ptr._version, scene_graph.future_vis.data_ptr(), scene_graph.future_vis._version, FT)
        ent = scene_graph.__dict__.get('_strive_batch_counts')
        if ent is not None and ent[0] == key:
            return ent[1]
        dev = scene_graph.future_vis.device
        sizes = (scene_graph.ptr[1:] - scene_graph.ptr[:-1]).to(device=dev, dtype=torch.float64)
        counts = torch.stack([(scene_graph.future_vis == 1.0).sum().to(torch.float64), sizes.sum(), (sizes * sizes - sizes).sum(),
                              torch.tensor(float(sizes.numel() * FT), dtype=torch.float64, device=dev)])
        scene_graph.__dict__['_strive_batch_counts'] = (key, counts)
        return counts

    def global_loss_share(self, loss_dict, local, glob):
        """This rank's additive share of the global-batch loss from the per-element outputs of TrafficModelLoss.  ``local`` /
        ``glob``: the four counts as device tensors (or sequences of floats); everything stays on the device."""
        w = self.loss_fn.loss_weights
        loss = w['recon'] * loss_dict['recon_loss'].sum() / glob[0] + w['kl'] * loss_dict['kl_loss'].sum() / glob[1]
        if 'coll_veh_prior' in loss_dict:
            cv = loss_dict['coll_veh_prior'].sum() * (local[2] / glob[2])
            if torch.is_tensor(local[2]):
                # a rank without pairs (one-agent scenes) has 0/0 = NaN locally: it contributes nothing
                cv = torch.where(local[2] > 0, cv, torch.zeros_like(cv))
            elif not local[2] > 0:
                cv = None
            if cv is not None:
                loss = loss + w['coll_veh_prior'] * cv
        if 'coll_env_prior' in loss_dict:
            loss = loss + w['coll_env_prior'] * loss_dict['coll_env_prior'].sum() / glob[3]
        return loss

    def step(self, scene_graph, map_idx, map_env, future_sample=None):
        dev = self.params[0].device
        local = self.batch_counts(scene_graph, self.model.FT).to(dev)
        glob = torch.clamp(self._all_reduce(local.clone()), min=1.0)      # an empty global count never divides; stays on the device
        w = self.loss_fn.loss_weights
        if future_sample is None:
            future_sample = w['coll_veh_prior'] > 0.0 or w['coll_env_prior'] > 0.0
        self.bucket.zero_()
        self._bind_grads()
        failed, loss_dict, share, error = 0.0, None, None, None
        try:
            from . import ops, params
            # weight packs are rebuilt after every optimiser step and need max |w| of every tensor on the host (operand scales are
            # kernel arguments): taken from the read-back of the PREVIOUS step (plus the most an optimiser step can add), so the
            # host never waits for the step it has just enqueued
            lr = max(float(g_['lr']) for g_ in self.optimizer.param_groups)
            with params.lagged_absmax(slack=8.0 * lr):
                pred = self.model(scene_graph, map_idx, map_env, future_sample=future_sample)
            loss_dict = self.loss_fn(scene_graph, pred, map_idx=map_idx, map_env=map_env)
            share = self.global_loss_share(loss_dict, local.to(torch.float32), glob.to(torch.float32))
            # the HIP backward calls accumulate straight into the bucket (every p.grad is a view of it): no scratch, no
            # per-parameter adds by autograd
            with ops.GradSink(self.params, self.bucket):
                share.backward()
        except Exception as e:             # ANY failure on one rank must still reach the collective below, or the others hang
            error = e
            self.last_error = e
            failed = 1.0
        self._bind_grads()                 # (a failed backward may have left some p.grad detached from the bucket)
        if failed:
            self.bucket.zero_()
        self.bucket[-1:].fill_(failed)     # (a fill kernel; `bucket[-1] = failed` is a synchronous copy of a host scalar)
        self._all_reduce(self.bucket)
        if error is not None and not isinstance(error, RuntimeError):
            raise error                    # the reference's loop only swallows RuntimeError (train_traffic.py:120-131)
        any_failed = failed
        if self._needs_host_vote():
            if self.vote_transport.startswith('host collective'):
                vote = torch.tensor([failed], dtype=torch.float32)              # a CPU tensor: the vote is a host collective
                dist.all_reduce(vote, op=dist.ReduceOp.SUM, group=self.vote_group)
                any_failed = float(vote[0])
            else:
                any_failed = float(self.bucket[-1])                             # (fallback: the flag summed with the gradients)
        if any_failed:
            return None                    # this rank or another one failed: nobody steps
        self.optimizer.step()
        # (fused optimisers update the parameters without bumping their version counters, which the pack caches key on)
        params.parameters_changed()
        out = dict(loss_dict)
        out['global_loss'] = self._all_reduce(share.detach().clone().reshape(1))
        return out

    def _needs_host_vote(self):
        """With more than one rank another rank may have failed while this one did not: the optimiser step must then be
        skipped here too.  The flags travel host to host (``vote_group``); a single rank knows its own outcome already."""
        return self._vote
