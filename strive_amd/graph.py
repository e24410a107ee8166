"""Minimal scene-graph containers: the subset of torch_geometric's Data/Batch the hot path uses.

The reference's drivers import ``torch_geometric.data.{Data,Batch,DataLoader}``
(reference src/adv_scenario_gen.py:13-14, src/refine_traffic_optim.py:16-17); PyG is not
installed here or on the GPU box, and only its collation semantics matter to the hot path:
concatenate node tensors, offset ``edge_index`` per graph, build ``ptr``/``batch``
(SURVEY.md §8(b)).  Scenes are always per-scene cliques without self loops, ego first
(reference src/datasets/nuscenes_dataset.py:678-687).
"""
import torch


def clique_edge_index(n):
    """(2, n(n-1)) long; row 0 = source j, row 1 = target i, source-major like the reference's
    itertools.product construction (reference src/datasets/nuscenes_dataset.py:681-687)."""
    if n <= 1:
        return torch.zeros((2, 0), dtype=torch.long)
    src = torch.arange(n).view(n, 1).expand(n, n)
    dst = torch.arange(n).view(1, n).expand(n, n)
    keep = src != dst
    return torch.stack([src[keep], dst[keep]], dim=0).contiguous()


class Data(object):
    """Attribute bag with ``in`` support, ``.to`` and key listing."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def __contains__(self, k):
        return k in self.__dict__ and self.__dict__[k] is not None

    def __getitem__(self, k):
        return self.__dict__[k]

    def __setitem__(self, k, v):
        self.__dict__[k] = v

    def to(self, device):
        for k in self.keys():
            v = self.__dict__[k]
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        out = self.__class__()
        for k in self.keys():
            v = self.__dict__[k]
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
        return out


class Batch(Data):
    """Collated scenes: node tensors concatenated on dim 0, ``edge_index`` offset by the running
    node count, ``ptr (B+1,)``, ``batch (NA,)``, ``num_graphs``."""

    NODE_KEYS = ('x', 'pos', 'past', 'past_gt', 'future', 'future_gt', 'sem', 'lw', 'past_vis', 'future_vis')

    @classmethod
    def from_data_list(cls, data_list):
        out = cls()
        sizes = [int(d.past.shape[0]) for d in data_list]
        keys = [k for k in data_list[0].keys()]
        for k in keys:
            vals = [d.__dict__[k] for d in data_list]
            if k == 'edge_index':
                offs, acc = [], 0
                for v, n in zip(vals, sizes):
                    offs.append(v + acc)
                    acc += n
                out.edge_index = torch.cat(offs, dim=1)
            elif torch.is_tensor(vals[0]):
                out.__dict__[k] = torch.cat(vals, dim=0)
            else:
                out.__dict__[k] = vals
        ptr = [0]
        for n in sizes:
            ptr.append(ptr[-1] + n)
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
        out.num_graphs = len(sizes)
        return out

    def to_data_list(self):
        res = []
        ptr = self.ptr.tolist()
        e_src = self.edge_index[0]
        for b in range(self.num_graphs):
            lo, hi = ptr[b], ptr[b + 1]
            d = Data()
            for k in self.keys():
                v = self.__dict__[k]
                if k in ('ptr', 'batch', 'num_graphs'):
                    continue
                if k == 'edge_index':
                    m = (e_src >= lo) & (e_src < hi)
                    d.edge_index = v[:, m] - lo
                elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == ptr[-1]:
                    d.__dict__[k] = v[lo:hi]
                else:
                    d.__dict__[k] = v
            res.append(d)
        return res
