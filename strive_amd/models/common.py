"""MLP container and bicycle step with the reference's names (reference src/models/common.py).

``MLP`` keeps the reference's parameter layout (``net.0`` Linear, ``net.{3k+1}`` LayerNorm,
``net.{3k+2}`` ReLU, ``net.{3k+3}`` Linear) so checkpoints load unchanged; its forward runs the fused
HIP kernel (strive_mlp_fwd); with autograd enabled the call is an autograd Function whose backward (strive_mlp_bwd) gives the
gradients w.r.t. the input and -- the training step -- all weights and biases (tests/test_training.py: all 174 parameter
gradients against the oracle's autograd).
"""
import torch
from torch import nn

from .. import ops


class MLP(nn.Module):
    def __init__(self, layers, nonlinearity=nn.ReLU, use_norm=True):
        """:param layers: list of layer sizes (including input/output)"""
        super(MLP, self).__init__()
        if nonlinearity is not nn.ReLU or not use_norm:
            raise NotImplementedError('the HIP MLP kernels implement Linear -> LayerNorm -> ReLU stacks only')
        mods = [nn.Linear(layers[0], layers[1])]
        for i in range(1, len(layers) - 1):
            mods.extend([nn.LayerNorm(layers[i]), nn.ReLU(), nn.Linear(layers[i], layers[i + 1])])
        self.net = nn.ModuleList(mods)
        self.layer_sizes = list(layers)

    def forward(self, x):
        return ops.mlp_forward(self, x)


def car_dynamics(kinematics, a, ddh, dt, xix, yix, hix, six, hdotix, vehicle_length, max_hdot, max_s):
    """Kinematic bicycle step on (B, N, 5) states (x, y, h, s, hdot); same signature and clamps as the
    reference (src/models/common.py:47-68).  Plain torch: it is only used outside the fused rollout."""
    newhdot = (kinematics[:, :, hdotix] + ddh * dt).clamp(-max_hdot, max_hdot)
    newh = kinematics[:, :, hix] + dt * kinematics[:, :, six].abs() / vehicle_length * newhdot
    news = (kinematics[:, :, six] + a * dt).clamp(0.0, max_s)
    newy = kinematics[:, :, yix] + news * newh.sin() * dt
    newx = kinematics[:, :, xix] + news * newh.cos() * dt
    cols = [None] * kinematics.shape[-1]
    cols[xix], cols[yix], cols[hix], cols[six], cols[hdotix] = newx, newy, newh, news, newhdot
    return torch.stack(cols, dim=-1)
