"""Scene interaction network with the reference's names and parameter layout
(reference src/models/interaction_net.py:16-218), evaluated by the fused HIP message-passing kernels.

Only the configuration TrafficModel instantiates is supported: one round (k=1), MLP update, max
aggregation over per-scene cliques.  ``AgentInteractionConv`` is a plain parameter container here --
torch_geometric is not a dependency.
"""
from torch import nn

from .common import MLP
from .. import ops


class AgentInteractionConv(nn.Module):
    def __init__(self, in_node_channels, in_sem_channels, in_edge_channels, out_channels, hidden_size=128,
                 gru_update=False, gru_single_step=False, nonlinearity=nn.ReLU, aggr='max'):
        super(AgentInteractionConv, self).__init__()
        if gru_update or gru_single_step or aggr != 'max' or hidden_size != 128:
            raise NotImplementedError('HIP message passing supports aggr="max", MLP update, hidden 128 only')
        self.edge_mlp = MLP([2 * (in_node_channels + in_sem_channels) + in_edge_channels, hidden_size, hidden_size,
                             out_channels], nonlinearity=nonlinearity)
        self.update_mlp = MLP([in_node_channels + out_channels + in_sem_channels, hidden_size, out_channels],
                              nonlinearity=nonlinearity)
        self.out_channels = out_channels


class SceneInteractionNet(nn.Module):
    def __init__(self, in_node_channels, in_sem_channels, in_edge_channels, msg_node_channels, out_channels,
                 gru_update=False, gru_single_step=False, k=1, nonlinearity=nn.ReLU):
        super(SceneInteractionNet, self).__init__()
        if gru_update or gru_single_step or k != 1 or in_edge_channels != 4:
            raise NotImplementedError('HIP message passing supports k=1, MLP update, 4-d edge poses only')
        self.mlp_in = MLP([in_node_channels, 128, 128, msg_node_channels], nonlinearity=nonlinearity)
        self.msg = nn.ModuleList([AgentInteractionConv(msg_node_channels, in_sem_channels, in_edge_channels,
                                                       msg_node_channels, hidden_size=128, nonlinearity=nonlinearity)])
        self.mlp_out = MLP([msg_node_channels, 128, 128, out_channels], nonlinearity=nonlinearity)
        self.NC = in_sem_channels

    def forward(self, scene_graph, h=None, return_out=True):
        """scene_graph: .x (NA,F) or (NA,NS,F), .pos, .sem, .edge_index (per-scene cliques), .ptr"""
        if h is not None or not return_out:
            raise NotImplementedError('hidden-state message passing is not used by TrafficModel')
        return ops.gnn_forward(self, scene_graph)
