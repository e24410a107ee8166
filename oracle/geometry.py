"""ORACLE (test infrastructure only) -- rigid transforms, bicycle dynamics, normalisers.

CPU restatement, in plain torch fp32, of the small geometric operators on STRIVE's
latent-optimisation hot path.  Each function cites the reference lines it restates.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (strive_amd/) never does.
"""
import torch


class Normalizer(object):
    """(x - mean) / std on the leading ``D`` components of the last dim.
    Restates MeanStdNormalizer (reference src/datasets/utils.py:44-113)."""

    def __init__(self, mean, std):
        self.mean_vals = mean.to(torch.float32)
        self.std_vals = std.to(torch.float32)

    def _ms(self, x):
        d = x.shape[-1]
        return self.mean_vals[:d].to(x.device), self.std_vals[:d].to(x.device)

    def normalize(self, x):
        m, s = self._ms(x)
        return (x - m) / s

    def unnormalize(self, x):
        m, s = self._ms(x)
        return x * s + m


def transform2frame(frame, poses, inverse=False):
    """Poses ``(B,N,4)`` of (x,y,hx,hy) into the frame ``(B,4)`` (or back out with ``inverse``).

    Restates reference src/utils/transforms.py:78-139 with the 2x2 products written out:
    forward   t' = R_f (t - t_f),  h' = first column of R_p R_f
    inverse   t' = R_f^T t + t_f,  h' = first column of R_p R_f^T
    with R_f = [[c, s], [-s, c]], R_p = [[pc, -ps], [ps, pc]].  Headings are not renormalised.
    """
    # The 2x2 products go through torch.matmul on (B,N,2,2) operands, as the reference does, so that the
    # CPU oracle reproduces the reference's rounding (batched-matmul inner products) bit for bit; the rollout
    # re-samples the map raster at every step, and a 1-ulp pose difference can flip a crop pixel.
    B, N, _ = poses.shape
    c, s = frame[:, 2], frame[:, 3]
    Rf = torch.stack([c, s, -s, c], dim=1).reshape(B, 1, 2, 2).expand(B, N, 2, 2)
    pc, ps = poses[:, :, 2], poses[:, :, 3]
    Rp = torch.stack([pc, -ps, ps, pc], dim=2).reshape(B, N, 2, 2)
    ft = frame[:, :2].reshape(B, 1, 2)
    if inverse:
        R = torch.matmul(Rp, Rf.transpose(2, 3))
        t = torch.matmul(Rf.transpose(2, 3), poses[:, :, :2].reshape(B, N, 2, 1))[:, :, :, 0] + ft
    else:
        R = torch.matmul(Rp, Rf)
        t = torch.matmul(Rf, (poses[:, :, :2] - ft).reshape(B, N, 2, 1))[:, :, :, 0]
    return torch.cat([t, torch.stack([R[:, :, 0, 0], R[:, :, 1, 0]], dim=2)], dim=-1)


def bicycle_step(state_u, a, ddh, veh_len, dt, max_hdot, max_s):
    """One kinematic-bicycle step on UNNORMALISED states ``(N,6)`` (x,y,hx,hy,s,hdot) with
    acceleration ``a (N,)``, yaw acceleration ``ddh (N,)`` and vehicle length ``veh_len (N,)``.

    Restates kinematics2angle -> car_dynamics -> kinematics2vec
    (reference src/utils/transforms.py:8-29, src/models/common.py:47-68,
    src/models/traffic_model.py:714-733) for a single step.
    """
    x, y, hx, hy, s, hdot = [state_u[:, i] for i in range(6)]
    h = torch.atan2(hy, hx)
    new_hdot = (hdot + ddh * dt).clamp(-max_hdot, max_hdot)
    new_h = h + dt * s.abs() / veh_len * new_hdot
    new_s = (s + a * dt).clamp(0.0, max_s)
    new_y = y + new_s * new_h.sin() * dt
    new_x = x + new_s * new_h.cos() * dt
    return torch.stack([new_x, new_y, new_h.cos(), new_h.sin(), new_s, new_hdot], dim=1)
