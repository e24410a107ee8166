"""Pose transforms with the reference's names (reference src/utils/transforms.py).  Plain torch: these
are host-side glue (encoder input assembly, loss bookkeeping); inside the fused kernels the same
arithmetic runs in registers (strive_amd/csrc/gnn_dev.h: rel_pose)."""
import torch

from ..ops import transform2frame as _t2f4


def kinematics2angle(kinematics):
    """(B,T,6) (x,y,hx,hy,s,hdot) -> (B,T,5) with heading angle.  (reference :8-17)"""
    h = torch.atan2(kinematics[:, :, 3:4], kinematics[:, :, 2:3])
    return torch.cat([kinematics[:, :, :2], h, kinematics[:, :, 4:]], dim=2)


def kinematics2vec(kinematics):
    """(B,T,5) (x,y,h,s,hdot) -> (B,T,6) with heading unit vector.  (reference :19-29)"""
    h = kinematics[:, :, 2]
    return torch.cat([kinematics[:, :, :2], torch.stack([h.cos(), h.sin()], dim=2), kinematics[:, :, 3:]], dim=-1)


def transform2frame(frame, poses, inverse=False):
    """Poses (B,N,3|4) into the local frame (B,3|4), or back with ``inverse``.  (reference :78-139)"""
    if poses.size(-1) == 4:
        return _t2f4(frame, poses, inverse=inverse)
    f4 = torch.stack([frame[:, 0], frame[:, 1], frame[:, 2].cos(), frame[:, 2].sin()], dim=1)
    p4 = torch.stack([poses[..., 0], poses[..., 1], poses[..., 2].cos(), poses[..., 2].sin()], dim=-1)
    o = _t2f4(f4, p4, inverse=inverse)
    return torch.cat([o[..., :2], torch.atan2(o[..., 3], o[..., 2]).unsqueeze(-1)], dim=-1)
