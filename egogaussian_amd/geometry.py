"""Host-side mirror of the reference's trainable object motion (/root/reference/utils/geometry_utils.py:14-88).

`ObjectMove` is what `GaussianModel.trainable_object_move` holds while an object pose is being estimated
(/root/reference/trainers/coarse_obj_pose.py, fine_obj.py): a translation and a 6-D rotation (Zhou et al., CVPR 2019).  With
`render(..., rot_cov=True, during_training=True)` the covariance of the object's Gaussians is built from
`rot_L(accum_R @ L)` (/root/reference/scene/gaussian_model.py:46-63), so the loss reaches `obj_rotation_6d` through the
covariance.  Same names, argument meaning and shapes as the reference; `rot_matrix()` is what this package's fused covariance
producer takes (fused.rotated_covariance_from_scaling_rotation(rot_matrix=...)).
"""
import torch
import torch.nn as nn
from torch.nn import functional as F


def matrix_to_rot6d(rotmat):
    """3x3 rotation -> its first two columns, shape (3, 2)  (geometry_utils.py:56-68)."""
    rot6d = torch.squeeze(rotmat.view(-1, 3, 3)[:, :, :2])
    assert rot6d.shape == (3, 2)
    return rot6d


def rot6d_to_matrix(rot_6d):
    """(3, 2) -> 3x3 by Gram-Schmidt on the two columns, third = their cross product  (geometry_utils.py:70-88)."""
    rot_6d = rot_6d.view(-1, 3, 2)
    a1, a2 = rot_6d[:, :, 0], rot_6d[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum("bi,bi->b", b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    rotmat = torch.squeeze(torch.stack((b1, b2, b3), dim=-1))
    assert rotmat.shape == (3, 3)
    return rotmat


def to_transform_mat(rot):
    M = torch.eye(4, device=rot.device, dtype=torch.float32)
    M[:3, :3] = rot.to(torch.float32)
    return M


class ObjectMove(nn.Module):
    def __init__(self):
        super().__init__()
        self.obj_translation = nn.Parameter(torch.zeros(3, dtype=torch.float))
        self.obj_rotation_6d = nn.Parameter(matrix_to_rot6d(torch.eye(3, dtype=torch.float)))

    def rot_matrix(self):
        return rot6d_to_matrix(self.obj_rotation_6d)

    def forward(self, xyz):
        if xyz.shape[1] == 3:
            xyz = torch.matmul(xyz, self.rot_matrix().t()) + self.obj_translation
        elif xyz.shape[1] == 4:                                   # the Gaussians' rotations, as the reference treats them
            xyz = torch.matmul(xyz, to_transform_mat(self.rot_matrix()).t())
        return xyz

    def rot_L(self, L):
        return torch.matmul(self.rot_matrix(), L)

    def capture(self):
        return self.obj_translation.detach(), rot6d_to_matrix(self.obj_rotation_6d.detach())

    def replace(self, new_trans, new_rot_6d):
        self.obj_translation = new_trans
        self.obj_rotation_6d = new_rot_6d
