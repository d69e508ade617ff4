"""Registry surface of the reference's `projects.mmdet3d_plugin` (ref: projects/mmdet3d_plugin/__init__.py:1-14):
importing this package registers every class a shipped config names."""
from . import bbox, dense, detector, extra_costs, head, losses, sparse_encoder, structures, transformer  # noqa: F401
from .bbox import (AssignResult, BBox3DL1Cost, FocalLossCost, HungarianAssigner3D, IoU3DCost, NMSFreeCoder, denormalize_bbox,
                   normalize_bbox)
from .dense import SECOND3D, SECOND3DFPN
from .detector import Uni3DETR
from .head import Uni3DETRHead
from .extra_costs import AxisAlignedIoU3DCost, RDIoUCost, RDIoULoss, RotatedIoU3DCost, SoftFocalLossCost, get_rdiou
from .losses import IoU3DLoss, L1Loss, SoftFocalLoss
from .sparse_encoder import SparseEncoderHD
from .structures import Boxes3D, DepthInstance3DBoxes, LiDARInstance3DBoxes
from .transformer import Uni3DETRTransformer, Uni3DETRTransformerDecoder, UniCrossAtten
