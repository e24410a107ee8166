"""Replay of one optimisation iteration as a HIP graph.

The shipped optimisation configs run ONE scene per batch (reference configs/adv_gen_rule_based.cfg:13,
configs/refine_traffic_optim.cfg:11: ``batch_size`` 1 / 10).  There an iteration -- zero_grad, rollout(s), loss, reverse
sweep, Adam -- is ~190 launches of a few microseconds each, and the host needs longer to enqueue them (1.6 ms) than a faster GPU
chain would need to run them.  The shapes of an iteration do not change over the 200-300 iterations of a loop (reference
src/utils/adv_gen_optim.py:106-175), so after a few eager iterations the whole iteration is captured once (torch.cuda.graph:
hipStreamBeginCapture on a side stream, the library's launches land in the capture because they are enqueued on torch's current
stream) and replayed with one hipGraphLaunch per iteration.

Requirements on the body: no host synchronisation (tools/sync_audit.py: the refine / adversarial / solution closures have
none), every tensor it reads lives across iterations, the optimiser is built with ``capturable=True`` (adam_kwargs()).
STRIVE_HIP_GRAPH = 0 | 1 | auto (default auto: iterations with ONE rollout and no logging callback, at any batch size --
round 5: the 64-agent limit of round 4 only encoded "no GPU gain above it"; the host side of a 512-agent refine iteration is
8.6 ms of Python per rank, which eight ranks on one host do not hide; STRIVE_HIP_GRAPH_MAX_AGENTS restores a limit)."""
import os

import torch

AUTO_MAX_AGENTS = int(os.environ.get('STRIVE_HIP_GRAPH_MAX_AGENTS', str(1 << 30)))


def graph_mode(n_agents, device, log=None, rollouts=1, closed_loop=False):
    """``rollouts`` = decoder rollouts per iteration (``closed_loop`` is informational: the closed loop's planner is capturable,
    its status look is skipped while capturing).  Iterations with TWO independent rollouts (adversarial, solution) are not
    replayed unless STRIVE_HIP_GRAPH=1 asks for it: eager, the two rollouts run on two HIP streams and their latency chains
    overlap (16 agents: 4.2 ms per iteration); a captured fork / join replays slowly on this runtime (8.6 ms) and a
    single-stream capture serialises the two chains (7.1 ms) -- profiles/r04_graph_ab.txt."""
    dev = torch.device(device)
    if dev.type != 'cuda' or not torch.cuda.is_available():
        return False
    env = os.environ.get('STRIVE_HIP_GRAPH', 'auto').lower()
    if env in ('0', 'off', 'false'):
        return False
    if log is not None:           # a logging callback reads loss entries on the host: that is a synchronisation per iteration
        return False
    if env in ('1', 'on', 'true'):
        return True
    return int(n_agents) <= AUTO_MAX_AGENTS and int(rollouts) == 1


def adam_kwargs(graphed):
    """torch.optim.Adam keeps its step count on the host unless told otherwise; a captured step needs it on the device."""
    return {'capturable': True} if graphed else {}


class GraphedIteration(object):
    """``body()`` = one whole iteration (zero_grad .. optimiser.step()) returning a tensor or a tuple of tensors.  Calls
    1 .. warmup run eagerly (lazy packs, cached tables and workspace buffers are built there), the next call captures the
    body and replays it, later calls replay.  The returned tensors are the capture's static outputs.  Scratch buffers the body
    requests while being captured belong to this object (ops.graph_workspaces), not to the process-wide cache.
    A capture that fails falls back to eager iterations -- same results -- and says so once (``fallback_reason``, a
    RuntimeWarning); anything that is not a capture failure (a shape error, a library error the eager body would raise as well)
    propagates."""

    def __init__(self, body, enabled, warmup=3):
        self.body, self.enabled, self.warmup = body, bool(enabled), int(warmup)
        self.calls, self.graph, self.out = 0, None, None
        self.replays = 0
        self.fallback_reason = None
        self._ws = {}

    @staticmethod
    def _is_capture_failure(e):
        msg = str(e).lower()
        return isinstance(e, RuntimeError) and any(k in msg for k in ('captur', 'hipgraph', 'cudagraph', 'graph', 'stream is'))

    def __call__(self):
        if not self.enabled:
            return self.body()
        self.calls += 1
        if self.calls <= self.warmup:
            return self.body()
        if self.graph is None:
            from .. import ops
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with ops.graph_workspaces(self._ws):
                    with torch.cuda.graph(g):
                        self.out = self.body()
            except RuntimeError as e:
                if not self._is_capture_failure(e):
                    raise
                # a body that cannot be captured (an unexpected synchronisation, an allocation the capture refuses) keeps running
                # eagerly: same results, but the caller is told -- the speed-up is gone and the failed attempt ran part of the
                # body's host side once more
                import warnings
                self.enabled = False
                self.fallback_reason = '%s: %s' % (type(e).__name__, str(e).splitlines()[0][:300])
                self._ws.clear()
                warnings.warn('HIP-graph capture of the optimisation iteration failed, continuing eagerly (%s)' % self.fallback_reason,
                              RuntimeWarning)
                torch.cuda.synchronize()
                return self.body()
            self.graph = g
        self.graph.replay()
        self.replays += 1
        return self.out
