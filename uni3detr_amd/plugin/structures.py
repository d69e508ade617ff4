"""Minimal stand-in for mmdet3d's box structures: only what the hot path reads — `.tensor` [G,7] with bottom-centre z
and `.gravity_center` (ref: projects/mmdet3d_plugin/models/dense_heads/uni3detr_head.py:759-761)."""
import torch


class Boxes3D:
    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        t = torch.as_tensor(tensor, dtype=torch.float32)
        if t.numel() == 0:
            t = t.reshape(0, box_dim)
        self.tensor = t.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - self.tensor.new_tensor(origin))
        self.box_dim = box_dim

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], dim=1)

    def to(self, *a, **k):
        b = Boxes3D.__new__(Boxes3D)
        b.tensor, b.box_dim = self.tensor.to(*a, **k), self.box_dim
        return b

    def __len__(self):
        return self.tensor.shape[0]


DepthInstance3DBoxes = LiDARInstance3DBoxes = Boxes3D
